import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import polars_ds_extension_amd as pds
from oracle import oracle as orc
p, bias = 8, True
rng = np.random.default_rng(100 + p)
G = 3000
sizes = rng.integers(1, 300, size=G)
sizes[::97] = rng.integers(0, p + 1, size=len(sizes[::97]))
off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
N = int(off[-1])
X = rng.normal(size=(N, p))
y = np.empty(N)
for g in range(G):
    s = slice(off[g], off[g + 1])
    y[s] = X[s] @ rng.normal(size=p) + 0.1 * rng.normal(size=sizes[g]) + (0.7 if bias else 0.0)
for g in range(5, G, 211):
    X[off[g] : off[g + 1], 1] = 2.0 * X[off[g] : off[g + 1], 0]
cols = [torch.from_numpy(np.ascontiguousarray(X[:, j])).cuda() for j in range(p)]
co, nu = pds.lin_reg_by(*cols, target=torch.from_numpy(y).cuda(), group_offsets=off, add_bias=bias)
co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
co_o, nu_o = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=bias, nthreads=4)
bad = np.flatnonzero(nu != nu_o)
print("null mismatches", len(bad), bad[:30], "sizes", sizes[bad[:30]], "gpu", nu[bad[:10]], "orc", nu_o[bad[:10]])
ok = ~nu & ~nu_o
err = np.linalg.norm(co[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)
idx = np.flatnonzero(ok)
w = idx[~(err < 1e-6)]
print("value mismatches", len(w), w[:40], "sizes", sizes[w[:40]])
print("groups mod 4 of bad:", np.bincount(w % 4, minlength=4), "nan rows:", np.isnan(co[ok]).any(axis=1).sum())
