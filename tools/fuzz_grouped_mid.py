"""Development aid: randomised shapes with 17 .. 64 features through the grouped paths (paired stream + row16 solver up to 32, record
stream + wave solver beyond) against the oracle: null decisions group by group, coefficients within 64 eps cond(X'X)."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import polars_ds_extension_amd as pds
from oracle import oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
F32 = os.environ.get("FUZZ_F32") == "1"   # f32 frames (17 .. 32 features take the paired stream): against the f64 truth of the same f32 data, 1e-4
if F32: pds.config.LIN_REG_EXPR_F64 = False
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 20)
n_cfg = 0; worst = 0.0; n_groups = n_null = 0
while time.time() < t_end:
    p = int(rng.integers(17, 33 if F32 else 65)); bias = bool(rng.integers(0, 2)); lam = float(rng.choice([0.0, 0.0, 0.3]))
    pp = p + bias
    G = int(rng.integers(1, 1500)); hi = int(rng.choice([pp + 3, 2 * pp, 6 * pp, 1500]))
    sizes = rng.integers(0, hi, size=G)
    if rng.integers(0, 3) == 0: sizes[rng.integers(0, G)] = int(rng.integers(5_000, 60_000))  # a group that spans waves
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    if N == 0: continue
    # (FUZZ_BIG_OFFSETS=1: a few columns with mean / sd of 30 .. 300 -- single pivot ratios of 1e3 .. 1e5 whose PRODUCT sits around the 1e12 gate)
    if os.environ.get("FUZZ_BIG_OFFSETS") == "1":
        X = rng.normal(size=(N, p))
        for j in rng.choice(p, size=int(rng.integers(1, 5)), replace=False): X[:, j] += float(rng.choice([30.0, 100.0, 300.0]))
    else:
        X = rng.normal(size=(N, p)) + rng.normal(size=p) * float(rng.choice([0.0, 0.3, 3.0]))
    y = X @ rng.normal(size=p) + rng.normal(size=N) * 0.1 + 0.5
    if F32: X, y = X.astype(np.float32), y.astype(np.float32)
    for g in rng.integers(0, G, size=G // 50):  # collinear / nearly collinear groups
        a, b = off[g], off[g + 1]
        if b - a > 2: X[a:b, 2] = X[a:b, 1] * 2.0 + (0.0 if rng.integers(0, 2) else 1e-6 * rng.normal(size=b - a))
    cols = [torch.from_numpy(np.ascontiguousarray(X[:, j])).cuda() for j in range(p)]
    tolkw = dict(singular_x_tol=1e-10) if F32 else {}   # (f32 frames: the same explicit gate on both sides, as tests/test_gpu_parity.py does)
    co, nu = pds.lin_reg_by(*cols, target=torch.from_numpy(y).cuda(), group_offsets=off, add_bias=bias, l2_reg=lam, **tolkw)
    co, nu = co.cpu().numpy().astype(np.float64), nu.cpu().numpy().astype(bool)
    if F32: X, y = X.astype(np.float64), y.astype(np.float64)
    co_o, nu_o = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=bias, l2_reg=lam, nthreads=16, **(dict(tol=1e-10) if F32 else {}))
    assert np.array_equal(nu, nu_o), (p, bias, lam, G, hi, np.flatnonzero(nu != nu_o)[:5], sizes[nu != nu_o][:5])
    ok = ~nu
    n_groups += G; n_null += int(nu.sum())
    if ok.any():
        err = np.linalg.norm(co[ok] - co_o[ok], axis=1) / np.maximum(np.linalg.norm(co_o[ok], axis=1), 1e-300)
        for g in np.flatnonzero(ok)[np.argsort(err)[-3:]]:
            Xg = X[off[g]: off[g + 1]]
            Xb = np.c_[Xg, np.ones(len(Xg))] if bias else Xg
            cnd = np.linalg.cond(Xb.T @ Xb + lam * np.eye(Xb.shape[1]))
            # (until round 6 f32 frames too small for the paired stream went through f32 moment records -- moments rounded to 6e-8; they take the
            #  paired stream now, the bound is kept)
            bound = max(1e-4, 2 * 6e-8 * cnd) if F32 else max(1e-10, 64 * 2.2e-16 * cnd)
            r = float(err[np.flatnonzero(ok) == g][0] / bound)
            worst = max(worst, r)
            assert r < 1.0, (p, bias, lam, g, int(sizes[g]), float(err.max()), bound)
    n_cfg += 1
print(f"{n_cfg} random configurations ok ({n_groups} groups, {n_null} null); worst error / (64 eps cond) on the worst groups {worst:.2e}")
