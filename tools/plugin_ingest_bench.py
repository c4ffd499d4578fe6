"""
Host-frame ingestion through the plugin boundary: wall time of `_polars_plugin_pl_lr` on an N x p Float64 frame that
lives in (pageable) host memory as single Arrow chunks -- what Polars hands over in select() context.  Separates the
marshalling cost (reference: series_to_slice_inner's memcpy, src/utils/mod.rs:101-206) from the H2D copy and the kernels.

  python tools/plugin_ingest_bench.py [n_rows] [n_feat] [path/to/lib.so | -] [sections: lr,out,by]
"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import pyarrow as pa

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import plugin_harness as ph  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import torch  # noqa: F401  (one HIP runtime per process, see _lib.load)

    sections = sys.argv[4].split(",") if len(sys.argv) > 4 else ["lr", "out", "by"]
    lib = C.CDLL(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "-" else __import__("polars_ds_extension_amd._lib", fromlist=["load"]).load()
    rng = np.random.default_rng(0)
    X = rng.random((p, n))
    y = X.T @ np.linspace(-1, 1, p) + 0.01 * rng.normal(size=n)
    ins = [("y", pa.array(y))] + [(f"x{j}", pa.array(X[j])) for j in range(p)]
    kw = {"bias": False, "null_policy": "raise", "l1_reg": 0.0, "l2_reg": 0.0, "solver": "qr", "tol": 1e-5, "max_iter": 200,
          "weighted": False, "positive": False, "singular_x_tol": 1e-12}
    gb = n * (p + 1) * 8 / 1e9
    for policy in ("raise", "skip") if "lr" in sections else ():
        kw["null_policy"] = policy
        ph.call_plugin(lib, "pl_lr", ins, kw)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            _, out = ph.call_plugin(lib, "pl_lr", ins, kw)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        print(f"pl_lr host frame {n} x {p} f64 ({gb:.2f} GB) null_policy={policy}: {t * 1e3:.1f} ms  = {gb / t:.1f} GB/s end to end", flush=True)
    # with a null in one column (bitmap import + device-side policy)
    mask = np.zeros(n, dtype=bool)
    mask[::1000] = True
    ins2 = [ins[0], ("x0", pa.array(X[0], mask=mask))] + ins[2:]
    kw["null_policy"] = "skip"
    ts = []
    for _ in range(4 if "lr" in sections else 0):
        t0 = time.perf_counter()
        ph.call_plugin(lib, "pl_lr", ins2, kw)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts[1:])) if ts else float("nan")
    print(f"  with nulls in one column: {t * 1e3:.1f} ms = {gb / t:.1f} GB/s", flush=True)

    # results as large as the inputs: rolling fit (N x p coefficients + pred back to the host) and pred / resid
    nr = min(n, 10_000_000)
    ins_r = [(nm, a.slice(0, nr)) for nm, a in ins]
    kr = {"null_policy": "raise", "n": 256, "bias": False, "lambda": 0.0, "min_size": p}
    gb_r = nr * ((p + 1) * 8 + (p + 1) * 8 + 8) / 1e9
    for sym, kwargs, gbs in (("pl_rolling_lr", kr, gb_r), ("pl_lr_pred", dict(kw, null_policy="raise"), nr * (p + 3) * 8 / 1e9)) if "out" in sections else ():
        ph.call_plugin(lib, sym, ins_r, kwargs)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            _, out = ph.call_plugin(lib, sym, ins_r, kwargs)
            ts.append(time.perf_counter() - t0)
            del out
        t = float(np.median(ts))
        print(f"{sym} host frame {nr} x {p}: {t * 1e3:.1f} ms = {gbs / t:.1f} GB/s (in + out bytes) end to end", flush=True)

    # key-aware grouped fit on a host frame: [key, y, x...] with 100 rows per key, keys ordered and shuffled
    ng = nr // 100
    key = np.repeat(np.arange(ng, dtype=np.int64), 100)
    for label, order in (("ordered keys", None), ("shuffled rows", rng.permutation(ng * 100))) if "by" in sections else ():
        cols = [("key", pa.array(key if order is None else key[order]))]
        cols += [(nm, pa.array(a.to_numpy()[: ng * 100] if order is None else a.to_numpy()[: ng * 100][order])) for nm, a in ins]
        kb = dict(kw, null_policy="raise")
        ph.call_plugin(lib, "pl_lr_by", cols, kb)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            _, out = ph.call_plugin(lib, "pl_lr_by", cols, kb)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        assert len(out) == ng
        print(f"pl_lr_by host frame {ng} groups x 100 rows x {p} ({label}): {t * 1e3:.1f} ms = {ng / t:.3g} regressions/s end to end", flush=True)


if __name__ == "__main__":
    main()
