"""
TEST INFRASTRUCTURE ONLY -- a stand-in for the device layer so that the HOST logic of csrc/plugin.cpp (Arrow import / export,
kwargs, null policies, key ordering, the coalescing queue) runs in the CPU suite.

`build()` compiles csrc/plugin.cpp (plain host C++) together with generated trampolines for every `pds_*` entry point of
include/pds_lstsq.h into tests/mock_device/_build/libpds_plugin_mock.so; `device.bind(lib)` routes the trampolines to Python
callbacks that answer with the CPU oracle (oracle/).  Nothing here is built by `__graft_entry__.build()`, imported by the
package, or loadable as the product library: the product path (libpds_lstsq_hip.so) has no CPU route and fails loudly
without an MI355X (tests/test_cabi_cpu.py::test_no_cpu_fallback_without_device).
"""
