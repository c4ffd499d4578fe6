"""
f32 parity, both distances (SURVEY.md 7 "f32 parity": compare GPU-f32 and the reference's all-f32 arithmetic to an f64 truth).
For every f32 entry point prints   d(gpu, truth)   d(oracle_f32, truth)   d(gpu, oracle_f32)
where truth = the oracle in f64 on the f32-rounded inputs.  Run on the GPU box:  python tools/f32_distances.py [big]
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import polars_ds_extension_amd as pds  # noqa: E402
from oracle import oracle as orc  # noqa: E402

pds.config.LIN_REG_EXPR_F64 = False
NT = min(64, orc.max_threads())


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def cols_of(X):
    return [dev(X[:, j]) for j in range(X.shape[1])]


def nrel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def rowrel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)))


def frel(a, b, floor=1e-9):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def line(name, gpu, o32, truth, d=nrel):
    print(f"{name:44s} gpu-truth {d(gpu, truth):9.2e}   orc32-truth {d(o32, truth):9.2e}   gpu-orc32 {d(gpu, o32):9.2e}", flush=True)


def make_xy(rng, n, p, noise=0.01):
    X = rng.random((n, p))
    beta = np.array([(-1.0) ** j * (0.05 + 0.03 * j) for j in range(p)])
    return X, X @ beta + noise * rng.normal(size=n)


rng = np.random.default_rng(21)
X, y = make_xy(rng, 400_000, 8, noise=0.05)
X32, y32 = X.astype(np.float32), y.astype(np.float32)
X64, y64 = X32.astype(np.float64), y32.astype(np.float64)
for bias in (False, True):
    line(f"ols 4e5x8 bias={bias}", pds.lin_reg(*cols_of(X32), target=dev(y32), add_bias=bias),
         orc.pl_lr(X32, y32, add_bias=bias, singular_x_tol=1e-6), orc.pl_lr(X64, y64, add_bias=bias))
line("ridge 0.1", pds.lin_reg(*cols_of(X32), target=dev(y32), l2_reg=0.1), orc.pl_lr(X32, y32, l2_reg=0.1, singular_x_tol=1e-6),
     orc.pl_lr(X64, y64, l2_reg=0.1))
kw = dict(l1_reg=0.001, l2_reg=0.001, tol=1e-7)
line("elastic net 4e5x8", pds.lin_reg(*cols_of(X32), target=dev(y32), **kw), orc.pl_lr(X32, y32, max_iter=2000, **kw),
     orc.pl_lr(X64, y64, l1_reg=0.001, l2_reg=0.001, tol=1e-9, max_iter=2000))
line("nnls", pds.lin_reg(*cols_of(X32), target=dev(y32), positive=True, tol=1e-7), orc.pl_lr(X32, y32, positive=True, tol=1e-7, max_iter=200),
     orc.pl_lr(X64, y64, positive=True, tol=1e-9, max_iter=2000))
for se in ("se", "hc0", "hc1", "hc2", "hc3"):
    r = pds.lin_reg_report(*cols_of(X32), target=dev(y32), add_bias=True, std_err=se)
    ro32 = orc.lin_reg_report(np.c_[X32, np.ones(len(y), np.float32)], y32, std_err=se)
    ro = orc.lin_reg_report(np.c_[X64, np.ones(len(y))], y64, std_err=se)
    key = "std_err" if se == "se" else f"{se}_se"
    line(f"report {se}: se", r[key], ro32["std_err"], ro["std_err"], frel)
    if se == "se":
        line("report: beta", r["beta"], ro32["beta"], ro["beta"])
        line("report: t", r["t"], ro32["t"], ro["t"], frel)
        line("report: r2", np.ravel(r["r2"])[:1], np.ravel(ro32["r2"])[:1], np.ravel(ro["r2"])[:1], frel)
off = np.arange(0, 400_001, 1000)
co, nu = pds.lin_reg_by(*cols_of(X32), target=dev(y32), group_offsets=off)
c32, _ = orc.grouped_lr([y32] + [X32[:, j] for j in range(8)], off, tol=1e-6, nthreads=NT)
c64, _ = orc.grouped_lr([y64] + [X64[:, j] for j in range(8)], off, nthreads=NT)
line("grouped 400x1000x8 (max over groups)", co.cpu().numpy(), c32, c64, rowrel)
off = np.arange(0, 400_001, 100)
co, nu = pds.lin_reg_by(*cols_of(X32), target=dev(y32), group_offsets=off, add_bias=True)
c32, _ = orc.grouped_lr([y32] + [X32[:, j] for j in range(8)], off, add_bias=True, tol=1e-6, nthreads=NT)
c64, _ = orc.grouped_lr([y64] + [X64[:, j] for j in range(8)], off, add_bias=True, nthreads=NT)
line("grouped 4000x100x8+bias (max over groups)", co.cpu().numpy(), c32, c64, rowrel)
Xn = rng.normal(size=(200_000, 8)).astype(np.float32)
yn = (Xn @ rng.normal(size=8) + 0.1 * rng.normal(size=200_000)).astype(np.float32)
off = np.arange(0, 200_001, 100)
co, nu = pds.lin_reg_by(*cols_of(Xn), target=dev(yn), group_offsets=off)
c32, _ = orc.grouped_lr([yn] + [Xn[:, j] for j in range(8)], off, tol=1e-6, nthreads=NT)
c64, _ = orc.grouped_lr([yn.astype(np.float64)] + [Xn[:, j].astype(np.float64) for j in range(8)], off, nthreads=NT)
line("grouped N(0,1) 2000x100x8 (max over groups)", co.cpu().numpy(), c32, c64, rowrel)
for w, p3 in ((64, 3), (256, 8)):
    n = 50_000
    co, pr, va = pds.rolling_lin_reg(*cols_of(X32[:n, :p3]), target=dev(y32[:n]), window_size=w)
    r32 = orc.rolling_lr(X32[:n, :p3], y32[:n], w)
    r64 = orc.rolling_lr(X64[:n, :p3], y64[:n], w)
    line(f"rolling w={w} p={p3} (max over rows)", co.cpu().numpy()[w - 1:], r32, r64, rowrel)
    direct = np.array([np.linalg.lstsq(X64[i - w + 1:i + 1, :p3], y64[i - w + 1:i + 1], rcond=None)[0] for i in range(w - 1, n, 997)])
    print(f"    vs direct f64 window solves: gpu {rowrel(co.cpu().numpy()[w - 1::997], direct):.2e}  orc32 {rowrel(r32[::997], direct):.2e}"
          f"  orc64 {rowrel(r64[::997], direct):.2e}")
n = 20_000
co, pr, va = pds.recursive_lin_reg(*cols_of(X32[:n, :3]), target=dev(y32[:n]), start_with=10)
r32 = orc.recursive_lr(X32[:n, :3], y32[:n], 10)
r64 = orc.recursive_lr(X64[:n, :3], y64[:n], 10)
line("recursive start=10 p=3 (max over rows)", co.cpu().numpy()[9:], r32, r64, rowrel)
# wide rolling (20 features + bias, w = 200)
rngw = np.random.default_rng(77)
n, p, w = 3000, 20, 200
Xw = rngw.random((n, p)).astype(np.float32)
yw = (Xw @ rngw.normal(size=p) + 0.3 + 0.05 * rngw.normal(size=n)).astype(np.float32)
co, pr, va = pds.rolling_lin_reg(*cols_of(Xw), target=dev(yw), window_size=w, add_bias=True)
Xwb = np.c_[Xw, np.ones(n, np.float32)]
r32 = orc.rolling_lr(Xwb, yw, w)
r64 = orc.rolling_lr(Xwb.astype(np.float64), yw.astype(np.float64), w)
line("wide rolling 20+bias w=200 (max over rows)", co.cpu().numpy()[w - 1:], r32, r64, rowrel)
# grouped lasso
G, per, q = 300, 60, 4
Xg = rngw.normal(size=(G * per, q)).astype(np.float32)
yg = (Xg @ rngw.normal(size=q) + 0.1 * rngw.normal(size=G * per)).astype(np.float32)
cl, _ = pds.lin_reg_by(*cols_of(Xg), target=dev(yg), group_offsets=np.arange(0, G * per + 1, per), l1_reg=0.01, tol=1e-7)
cl = cl.cpu().numpy()
o32 = np.array([orc.pl_lr(Xg[g * per:(g + 1) * per], yg[g * per:(g + 1) * per], l1_reg=0.01, tol=1e-7, max_iter=2000) for g in range(G)])
o64 = np.array([orc.pl_lr(Xg[g * per:(g + 1) * per].astype(np.float64), yg[g * per:(g + 1) * per].astype(np.float64), l1_reg=0.01, tol=1e-10,
                          max_iter=2000) for g in range(G)])
line("grouped lasso 300x60x4 (max over groups)", cl, o32, o64, rowrel)
# HC3 with 20 features; OLS with 130
pw = 70
Xh = rngw.normal(size=(900, pw)).astype(np.float32)
yh = (Xh @ rngw.normal(size=pw) + 0.1 * rngw.normal(size=900)).astype(np.float32)
r = pds.lin_reg_report(*cols_of(Xh[:, :20]), target=dev(yh), std_err="hc3")
line("report hc3 900x20", r["hc3_se"], orc.lin_reg_report(Xh[:, :20], yh, std_err="hc3")["std_err"],
     orc.lin_reg_report(Xh[:, :20].astype(np.float64), yh.astype(np.float64), std_err="hc3")["std_err"], frel)
cw, nw = pds.lin_reg_by(*cols_of(Xh), target=dev(yh), group_offsets=np.array([0, 400, 900]))
line("grouped >64 feats (group 1: 500x70)", cw.cpu().numpy()[1], orc.pl_lr(Xh[400:], yh[400:], singular_x_tol=1e-6),
     orc.pl_lr(Xh[400:].astype(np.float64), yh[400:].astype(np.float64)))

if len(sys.argv) > 1 and sys.argv[1] == "big":
    # configs[4] at >= 1e6 rows: elastic net, p = 512 f32, AR(0.5)-correlated columns, 32 non-zero coefficients
    rng = np.random.default_rng(4)
    n, p = 1_000_000, 512
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    E = torch.randn(n, p, dtype=torch.float32, device="cuda", generator=g)
    Xt = torch.empty(p, n, dtype=torch.float32, device="cuda")
    Xt[0] = E[:, 0]
    for j in range(1, p):
        Xt[j] = 0.5 * Xt[j - 1] + float(np.sqrt(0.75)) * E[:, j]
    del E
    beta = np.zeros(p)
    beta[rng.choice(p, 32, replace=False)] = rng.normal(size=32)
    yt = (Xt.t().double() @ torch.from_numpy(beta).cuda()).float() + 0.5 * torch.randn(n, dtype=torch.float32, device="cuda", generator=g)
    t0 = time.time()
    b = pds.lin_reg(*[Xt[j] for j in range(p)], target=yt, l1_reg=0.01, l2_reg=0.01, tol=1e-5)
    print(f"config5 1e6x512 gpu fit {time.time() - t0:.2f} s", flush=True)
    Xh = np.ascontiguousarray(Xt.t().cpu().numpy())
    yh = yt.cpu().numpy()
    t0 = time.time()
    o32 = orc.coordinate_descent(Xh, yh, 0.01, 0.01, False, 1e-5, 2000, False, nthreads=NT)
    print(f"   oracle f32 {time.time() - t0:.1f} s", flush=True)
    t0 = time.time()
    o64 = orc.coordinate_descent(Xh.astype(np.float64), yh.astype(np.float64), 0.01, 0.01, False, 1e-9, 2000, False, nthreads=NT)
    print(f"   oracle f64 {time.time() - t0:.1f} s", flush=True)
    line("config5 EN 1e6x512", b, o32, o64)
    print("   nnz gpu/orc32/orc64", int(np.sum(np.abs(b) > 1e-6)), int(np.sum(np.abs(o32) > 1e-6)), int(np.sum(np.abs(o64) > 1e-6)))
