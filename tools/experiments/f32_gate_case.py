"""Development aid: replays tools/fuzz_grouped_mid.py (FUZZ_F32=1) up to its first null-flag mismatch and prints what both sides see for
that group: rows, the f64 gate statistic of the f32 data (pivoted QR log-det rule, lr_solvers.rs:341-380) and who says null."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch, scipy.linalg
import polars_ds_extension_amd as pds
from oracle import oracle as orc
pds.config.LIN_REG_EXPR_F64 = False
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
for it in range(2000):
    p = int(rng.integers(17, 33)); bias = bool(rng.integers(0, 2)); lam = float(rng.choice([0.0, 0.0, 0.3]))
    pp = p + bias
    G = int(rng.integers(1, 1500)); hi = int(rng.choice([pp + 3, 2 * pp, 6 * pp, 1500]))
    sizes = rng.integers(0, hi, size=G)
    if rng.integers(0, 3) == 0: sizes[rng.integers(0, G)] = int(rng.integers(5_000, 60_000))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    if N == 0: continue
    X = rng.normal(size=(N, p)) + rng.normal(size=p) * float(rng.choice([0.0, 0.3, 3.0]))
    y = X @ rng.normal(size=p) + rng.normal(size=N) * 0.1 + 0.5
    X, y = X.astype(np.float32), y.astype(np.float32)
    for g in rng.integers(0, G, size=G // 50):
        a, b = off[g], off[g + 1]
        if b - a > 2: X[a:b, 2] = X[a:b, 1] * 2.0 + (0.0 if rng.integers(0, 2) else 1e-6 * rng.normal(size=b - a))
    cols = [torch.from_numpy(np.ascontiguousarray(X[:, j])).cuda() for j in range(p)]
    co, nu = pds.lin_reg_by(*cols, target=torch.from_numpy(y).cuda(), group_offsets=off, add_bias=bias, l2_reg=lam, singular_x_tol=1e-10)
    nu = nu.cpu().numpy().astype(bool)
    Xd, yd = X.astype(np.float64), y.astype(np.float64)
    co_o, nu_o = orc.grouped_lr([yd] + [Xd[:, j] for j in range(p)], off, add_bias=bias, l2_reg=lam, nthreads=16, tol=1e-10)
    bad = np.flatnonzero(nu != nu_o)
    if len(bad) == 0: continue
    print(f"config {it}: p={p} bias={bias} lam={lam} G={G}; mismatching groups {bad[:5]}")
    for g in bad[:3]:
        a, b = off[g], off[g + 1]
        Xg = np.c_[Xd[a:b], np.ones(b - a)] if bias else Xd[a:b]
        A = Xg.T @ Xg + lam * np.eye(Xg.shape[1])
        _, R, _ = scipy.linalg.qr(A, pivoting=True)
        stat = np.sum(np.log(np.abs(np.diag(R)))) - np.sum(np.log(np.diag(A)))
        col2 = Xd[a:b, 2] - 2.0 * Xd[a:b, 1]
        print(f"  group {g}: rows {b - a}, library null {bool(nu[g])}, oracle null {bool(nu_o[g])}; ln|det R| - sum ln A_ii = {stat:.6f} (ln tol = {np.log(1e-10):.6f}); "
              f"max |x2 - 2 x1| = {np.abs(col2).max():.3e}; cond(A) = {np.linalg.cond(A):.3e}")
    break
