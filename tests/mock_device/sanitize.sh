#!/bin/bash
# AddressSanitizer + UBSan, then ThreadSanitizer, over the host logic of csrc/plugin.cpp (mock device layer, CPU only).
#   bash tests/mock_device/sanitize.sh
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
B="$ROOT/tests/mock_device/_build"
python "$ROOT/tests/mock_device/build.py" > /dev/null
SRC="$ROOT/polars_ds_extension_amd/csrc/plugin.cpp $B/mock_capi.cpp"
g++ -O1 -g -std=c++17 -shared -fPIC -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -o "$B/libpds_plugin_mock_asan.so" $SRC
g++ -O1 -g -std=c++17 -shared -fPIC -pthread -fsanitize=thread -fno-omit-frame-pointer -o "$B/libpds_plugin_mock_tsan.so" $SRC
cd "$ROOT"
echo "== ASan + UBSan: Series tests"
PDS_MOCK_LIB="$B/libpds_plugin_mock_asan.so" LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_plugin_host_logic.py -x -q -p no:cacheprovider
echo "== TSan: coalescing queue under 2 ... 64 calling threads"
PDS_MOCK_LIB="$B/libpds_plugin_mock_tsan.so" LD_PRELOAD="$(gcc -print-file-name=libtsan.so)" TSAN_OPTIONS="halt_on_error=1 report_signal_unsafe=0" \
  python - <<'PY'
import ctypes as C, pickle, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from mock_device import device
lib = device.load()
kw = pickle.dumps({"bias": False, "null_policy": "raise", "l1_reg": 0.0, "l2_reg": 0.0, "solver": "qr", "tol": 1e-5, "max_iter": 200,
                   "weighted": False, "positive": False, "singular_x_tol": 1e-12}, protocol=5)
buf = (C.c_uint8 * len(kw)).from_buffer_copy(kw)
sec, dev = C.c_double(), C.c_double()
for threads, calls in ((2, 50), (8, 40), (32, 20), (64, 10)):
    fails = lib.pds_plugin_debug_concurrent_lr(threads, calls, 60, 3, buf, len(kw), C.byref(sec), C.byref(dev))
    b, r, m = C.c_longlong(), C.c_longlong(), C.c_longlong()
    lib.pds_plugin_debug_coalesce_stats(C.byref(b), C.byref(r), C.byref(m), 1)
    print(f"{threads} threads x {calls} calls: failures {fails}, max deviation {dev.value:.1e}, {r.value} requests in {b.value} batches (largest {m.value})")
    assert fails == 0 and dev.value == 0.0
PY
echo "sanitizers: clean"
