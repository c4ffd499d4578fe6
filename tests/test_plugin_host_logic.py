"""
The host logic of the Polars plugin layer (csrc/plugin.cpp: Arrow import / export, kwargs, null policies, key ordering,
result assembly, the coalescing queue) in the CPU suite: the SAME Series-in / Series-out tests that run against
libpds_lstsq_hip.so on an MI355X (tests/test_plugin_abi.py, `-m gpu`) run here against plugin.cpp linked to a mock device
layer that answers with the oracle (tests/mock_device -- test infrastructure, never part of the product library).
What these runs prove is the plumbing, not the kernels: the numbers come from the oracle on both sides.
"""
import inspect
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import test_plugin_abi as T  # noqa: E402

SERIES_TESTS = [name for name, fn in vars(T).items()
                if name.startswith("test_") and callable(fn) and any(m.name == "gpu" for m in getattr(fn, "pytestmark", []))]


@pytest.fixture(scope="module")
def mock_so(orc):
    from mock_device import device

    return device.load()


def test_every_series_test_is_covered():
    assert len(SERIES_TESTS) >= 11


@pytest.mark.parametrize("name", SERIES_TESTS)
def test_series_tests_against_the_mock_device(name, mock_so, orc, golden):
    fn = getattr(T, name)
    avail = {"so": mock_so, "orc": orc, "golden": golden}
    fn(**{k: avail[k] for k in inspect.signature(fn).parameters})


def test_mock_is_not_the_product_library(mock_so):
    from polars_ds_extension_amd import _lib

    import ctypes as C

    mock_so.pds_version.restype = C.c_char_p
    assert b"mock" in mock_so.pds_version()
    assert "mock_device" in str(Path(mock_so._name)) and Path(mock_so._name) != _lib.LIB_PATH
