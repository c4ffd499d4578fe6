"""Stress of the sliced multi-context by-key route against the single-context call: repeated runs, where do null flags / coefficients differ?"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import polars_ds_extension_amd as pds  # noqa: E402

rng = np.random.default_rng(5)
G, p = 60_000, 5
sizes = rng.integers(20, 120, size=G)
keys_g = np.cumsum(rng.integers(1, 4, size=G)) - 7
key = np.repeat(keys_g, sizes).astype(np.int64)
N = len(key)
X = rng.normal(size=(N, p))
y = X @ rng.normal(size=p) + 1e-3 * key + 0.1 * rng.normal(size=N)
off = np.concatenate([[0], np.cumsum(sizes)])
coll = list(range(11, G, 997))
for g in coll:
    X[off[g]: off[g + 1], 1] = 2.0 * X[off[g]: off[g + 1], 0]
cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
k1, c1, n1 = pds.lin_reg_by_key(*cols, target=y, key=key, add_bias=True)
print("single: nulls", int(n1.sum()), "expected", len(coll), "equal to collinear set:", set(np.flatnonzero(n1)) == set(coll))
for n_ctx, n_slices in ((2, 0), (2, 8), (3, 7), (4, 4), (1, 3), (2, 2)):
    ctxs = [pds.Context(0) for _ in range(n_ctx)]
    bad = 0
    for rep in range(6):
        k2, c2, n2 = pds.lin_reg_by_key_multi(*cols, target=y, key=key, contexts=ctxs, n_slices=n_slices, add_bias=True)
        d = np.flatnonzero(n1 != n2)
        ok = ~n1.astype(bool) & ~n2.astype(bool)
        dev = np.linalg.norm(c1[ok] - c2[ok], axis=1) / np.linalg.norm(c1[ok], axis=1)
        if len(d) or dev.max() > 1e-11 or not np.array_equal(k1, k2):
            bad += 1
            print(f"  n_ctx {n_ctx} slices {n_slices} rep {rep}: {len(d)} flag differences at groups {d[:8]} (single {n1[d[:8]]}, multi {n2[d[:8]]}), "
                  f"in collinear set {[int(g) in coll for g in d[:8]]}; max coeff dev {dev.max():.2e} at {np.flatnonzero(ok)[dev.argmax()]}")
    print(f"n_ctx {n_ctx} slices {n_slices}: {bad} of 6 runs differ")
    for c in ctxs:
        c.close()
