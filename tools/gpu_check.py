"""Quick on-GPU sanity script (development aid): prints max relative differences vs the CPU oracle."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import polars_ds_extension_amd as pds
from polars_ds_extension_amd import config
from oracle import oracle as orc

def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))

def nrm(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

print("torch", torch.__version__, torch.cuda.get_device_name(0))
ctx = pds.default_context()
print("CUs", ctx.num_cus)
rng = np.random.default_rng(0)
for dt in (np.float64, np.float32):
    config.LIN_REG_EXPR_F64 = dt == np.float64
    for n, p in ((1000, 4), (100_003, 16), (5, 2), (129, 1), (1_000_000, 7)):
        X = rng.random((n, p)); beta = rng.normal(size=p); y = X @ beta + 0.01 * rng.normal(size=n)
        Xd, yd = X.astype(dt), y.astype(dt)
        cols = [np.ascontiguousarray(Xd[:, j]) for j in range(p)]
        M = pds.gram_moments(*cols, target=yd)
        Z = np.c_[Xd.astype(np.float64), np.ones(n), yd.astype(np.float64)]
        Mref = Z.T @ Z
        tcols = [torch.from_numpy(c).cuda() for c in cols]; ty = torch.from_numpy(yd).cuda()
        M2 = pds.gram_moments(*tcols, target=ty)
        for bias in (False, True):
            b = pds.lin_reg(*tcols, target=ty, add_bias=bias)
            bo = orc.pl_lr(Xd, yd, add_bias=bias, singular_x_tol=1e-12 if dt == np.float64 else 1e-6)
            print(f"{dt.__name__} n={n} p={p} bias={bias}: moments host {nrm(M, Mref):.2e} dev {nrm(M2, Mref):.2e}  coeff rel {rel(b, bo) if b is not None and bo is not None else (b, bo)}")
config.LIN_REG_EXPR_F64 = True
n, p = 200_000, 8
X = rng.random((n, p)); y = X @ rng.normal(size=p) + 0.5 + 0.1 * rng.normal(size=n)
cols = [torch.from_numpy(np.ascontiguousarray(X[:, j])).cuda() for j in range(p)]; ty = torch.from_numpy(y).cuda()
for kw in (dict(l2_reg=0.1), dict(l1_reg=0.01), dict(l1_reg=0.01, l2_reg=0.02), dict(positive=True), dict(solver="choleskey"), dict(solver="svd"), dict(l2_reg=0.3, positive=True)):
    for bias in (False, True):
        b = pds.lin_reg(*cols, target=ty, add_bias=bias, tol=1e-9, max_iter=2000, **kw)
        bo = orc.pl_lr(X, y, add_bias=bias, tol=1e-9, max_iter=2000, **kw)
        print(kw, bias, "rel", rel(b, bo), "abs", float(np.max(np.abs(b - bo))))
w = rng.random(n) + 0.1
b = pds.lin_reg(*cols, target=ty, add_bias=True, weights=torch.from_numpy(w).cuda())
print("weighted", rel(b, orc.pl_lr(X, y, add_bias=True, weights=w)))
pr = pds.lin_reg(*cols, target=ty, add_bias=True, return_pred=True)
bo = orc.pl_lr(X, y, add_bias=True)
print("pred", nrm(pr[0].cpu().numpy(), np.c_[X, np.ones(n)] @ bo), "resid", nrm(pr[1].cpu().numpy(), y - np.c_[X, np.ones(n)] @ bo))
for se in ("se", "hc0", "hc1", "hc2", "hc3"):
    r = pds.lin_reg_report(*cols, target=ty, add_bias=True, std_err=se, y_var=float(np.var(y, ddof=1)))
    ro = orc.lin_reg_report(np.c_[X, np.ones(n)], y, std_err=se)
    k = [k for k in r if k.endswith("se") or k == "std_err"][0]
    print(se, "beta", rel(r["beta"], ro["beta"]), "se", rel(r[k], ro["std_err"]), "t", rel(r["t"], ro["t"]), "p", rel(r["p>|t|"], np.maximum(ro["p"], 1e-300)), "ci", rel(r["0.025"], ro["ci_lo"]), "r2", abs(r["r2"][0] - ro["r2"]), abs(r["adj_r2"][0] - ro["adj_r2"]))
r = pds.lin_reg_report(*cols, target=ty, add_bias=True, weights=torch.from_numpy(w).cuda(), y_var=float(np.var(y, ddof=1)))
ro = orc.wls_report(np.c_[X, np.ones(n)], y, w)
print("wls", rel(r["beta"], ro["beta"]), rel(r["std_err"], ro["std_err"]), rel(r["p>|t|"], np.maximum(ro["p"], 1e-300)))
# grouped
G = 5000
sizes = rng.integers(9, 260, size=G); off = np.concatenate([[0], np.cumsum(sizes)]); N = int(off[-1]); p = 8
X = rng.normal(size=(N, p)); y = np.empty(N)
for g in range(G):
    s = slice(off[g], off[g + 1]); y[s] = X[s] @ rng.normal(size=p) + 0.1 * rng.normal(size=sizes[g])
for g in range(0, G, 500):  # collinear groups -> null
    X[off[g]:off[g + 1], 1] = 2 * X[off[g]:off[g + 1], 0]
cols = [torch.from_numpy(np.ascontiguousarray(X[:, j])).cuda() for j in range(p)]; ty = torch.from_numpy(y).cuda()
for bias in (False, True):
    co, nu = pds.lin_reg_by(*cols, target=ty, group_offsets=off, add_bias=bias)
    co = co.cpu().numpy(); nu = nu.cpu().numpy()
    worst = 0.0; nmis = 0
    for g in range(G):
        s = slice(off[g], off[g + 1])
        bo = orc.pl_lr(X[s], y[s], add_bias=bias) if sizes[g] >= p + bias else None
        if (bo is None) != bool(nu[g]): nmis += 1
        elif bo is not None: worst = max(worst, nrm(co[g], bo))
    print("grouped bias", bias, "null mismatches", nmis, "nulls", int(nu.sum()), "worst normwise", worst)
t0 = time.time()
try:
    co, pr, va = pds.rolling_lin_reg(*cols[:3], target=ty, window_size=16)
    print("rolling ran", time.time() - t0)
except Exception as e:
    print("rolling:", e)
