"""
Development aid / measurement: the per-group call pattern of an unchanged `df.group_by(key).agg(pds.lin_reg(...))` --
T host threads (Polars' rayon pool) each calling `_polars_plugin_pl_lr` on ~100-row frames -- with and without the
plugin layer's coalescing queue (PDS_PLUGIN_COALESCE=0 in a fresh process switches it off).
Usage: python tools/coalesce_bench.py [threads] [calls_per_thread] [rows] [features]
"""
import ctypes as C, os, pickle, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
T, K, N, P = (int(a) for a in (sys.argv[1:5] + ["32", "400", "100", "8"][len(sys.argv) - 1:]))
if os.environ.get("_PDS_COALESCE_CHILD"):
    from polars_ds_extension_amd import _lib
    so = _lib.load()
    kw = pickle.dumps({"bias": False, "null_policy": "raise", "l1_reg": 0.0, "l2_reg": 0.0, "solver": "qr", "tol": 1e-5,
                       "max_iter": 200, "weighted": False, "positive": False, "singular_x_tol": 1e-12}, protocol=5)
    buf = (C.c_uint8 * len(kw)).from_buffer_copy(kw)
    sec, dev = C.c_double(), C.c_double()
    so.pds_plugin_debug_concurrent_lr(4, 20, N, P, buf, len(kw), C.byref(sec), C.byref(dev))  # warm-up: contexts, code objects
    so.pds_plugin_debug_coalesce_stats(None, None, None, 1)
    fails = so.pds_plugin_debug_concurrent_lr(T, K, N, P, buf, len(kw), C.byref(sec), C.byref(dev))
    b, r, m = C.c_longlong(), C.c_longlong(), C.c_longlong()
    so.pds_plugin_debug_coalesce_stats(C.byref(b), C.byref(r), C.byref(m), 0)
    print(f"coalesce={os.environ.get('PDS_PLUGIN_COALESCE', '1')}: {T} threads x {K} calls of {N} x {P}: {sec.value * 1e3:.1f} ms "
          f"= {T * K / sec.value:,.0f} regressions/s ({sec.value / K * 1e6:.1f} us per call per thread); failures {fails}; "
          f"max deviation between repeats {dev.value:.2e}; batches {b.value}, requests {r.value}, largest batch {m.value}")
else:
    for mode in os.environ.get("MODES", "1,0").split(","):
        env = dict(os.environ, _PDS_COALESCE_CHILD="1", PDS_PLUGIN_COALESCE=mode)
        subprocess.run([sys.executable, __file__, str(T), str(K), str(N), str(P)], env=env, check=False)
