import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import polars_ds_extension_amd as pds
n_ctx, n_slices = 1, 3
rng = np.random.default_rng(500 + 10 * n_ctx + n_slices)
G, p = 60_000, 5
sizes = rng.integers(20, 120, size=G)
keys_g = np.cumsum(rng.integers(1, 4, size=G)) - 7
key = np.repeat(keys_g, sizes).astype(np.int64)
N = len(key)
X = rng.normal(size=(N, p))
y = X @ rng.normal(size=p) + 1e-3 * key + 0.1 * rng.normal(size=N)
off = np.concatenate([[0], np.cumsum(sizes)])
for g in range(11, G, 997):
    X[off[g]: off[g + 1], 1] = 2.0 * X[off[g]: off[g + 1], 0]
cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
ctxs = [pds.Context(0) for _ in range(n_ctx)]
k1, c1, n1 = pds.lin_reg_by_key(*cols, target=y, key=key, add_bias=True)
gid = np.repeat(np.arange(G), sizes)
for name, fn in (("single", lambda: pds.lin_reg_by_key_pred(*cols, target=y, key=key, add_bias=True)),
                 ("multi", lambda: pds.lin_reg_by_key_pred_multi(*cols, target=y, key=key, contexts=ctxs, n_slices=n_slices, add_bias=True))):
    pr, rs, f = fn()
    bad = np.isnan(pr) & (f == 0)
    print(name, "nan preds in unflagged rows:", bad.sum(), "flagged rows:", (f != 0).sum(), "flag values:", np.unique(f))
    if bad.any():
        gb = np.unique(gid[bad])
        print("  groups:", gb[:10], "n1 of those:", n1[gb[:10]], "coef:", c1[gb[0]], "rows in group", sizes[gb[0]], "bad rows in it", bad[gid == gb[0]].sum())
        r = np.flatnonzero(bad)[:10]
        print("  rows:", r, "row mod 128:", r % 128)
