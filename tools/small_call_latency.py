"""Per-call latency of the host-buffer entry points on small frames (what Polars pays when it calls pl_lr once per group)."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import polars_ds_extension_amd as pds
from polars_ds_extension_amd import _lib
import ctypes as C

rng = np.random.default_rng(0)
out = {}
ctx = pds.Context(0)
for n, p in ((100, 8), (1000, 8), (10000, 16)):
    X = rng.random((n, p)); y = X @ rng.normal(size=p) + 0.1 * rng.normal(size=n)
    xs = [np.ascontiguousarray(X[:, j]) for j in range(p)]
    for _ in range(20): pds.lin_reg(*xs, target=y, add_bias=True, ctx=ctx)
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps): pds.lin_reg(*xs, target=y, add_bias=True, ctx=ctx)
    dt = (time.perf_counter() - t0) / reps
    # raw C ABI call, arguments prepared once (no Python marshalling in the loop)
    lib = _lib.load()
    cols = (C.c_void_p * (p + 1))(y.ctypes.data, *[x.ctypes.data for x in xs])
    prm = _lib.LRParams(add_bias=1, l1_reg=0.0, l2_reg=0.0, tol=1e-5, solver=0, positive=0, max_iter=200, singular_x_tol=1e-12)
    co = np.empty(p + 1); isn = C.c_int(0)
    f = lib.pds_lr_f64
    for _ in range(20): f(ctx._h, cols, None, p, C.c_int64(n), 0, C.byref(prm), C.c_void_p(co.ctypes.data), C.byref(isn))
    t0 = time.perf_counter()
    for _ in range(reps): f(ctx._h, cols, None, p, C.c_int64(n), 0, C.byref(prm), C.c_void_p(co.ctypes.data), C.byref(isn))
    dc = (time.perf_counter() - t0) / reps
    out[f"{n}x{p}"] = {"python_us": round(dt * 1e6, 1), "c_abi_us": round(dc * 1e6, 1)}
print(json.dumps(out, indent=1))
