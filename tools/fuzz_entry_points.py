"""Development aid: random shapes through lin_reg (all methods), lin_reg_report, rolling / recursive fits and the keyed grouped
path against the oracle -- run on the GPU box after kernel changes.  Usage: python tools/fuzz_entry_points.py [seed] [seconds]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import polars_ds_extension_amd as pds
from oracle import oracle as orc

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
nrel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-300))
count = {}
worst = {}
def note(k, e):
    count[k] = count.get(k, 0) + 1
    worst[k] = max(worst.get(k, 0.0), e)

while time.time() < t_end:
    kind = rng.choice(["lr", "lr", "report", "rolling", "recursive", "keyed", "weighted", "multi", "grouped", "f32"])
    if kind in ("lr", "weighted", "multi"):
        p = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 24, 40, 70]))
        n = int(rng.integers(max(3 * p, 40), 60_000))
        bias = bool(rng.integers(0, 2))
        X = rng.normal(size=(n, p)) + rng.normal(size=p) * float(rng.choice([0.0, 2.0]))
        y = X @ rng.normal(size=p) + 0.3 + 0.1 * rng.normal(size=n)
        cols = [dev(X[:, j]) for j in range(p)]
        if kind == "weighted":
            w = rng.random(n) + 0.1
            b = pds.lin_reg(*cols, target=dev(y), add_bias=bias, weights=dev(w))
            note("weighted", nrel(b, orc.pl_lr(X, y, add_bias=bias, weights=w)))
            assert worst["weighted"] < 1e-9, (p, n, bias)
            continue
        if kind == "multi":
            k = int(rng.integers(2, 5))
            Y = np.c_[[y + 0.5 * j * X[:, 0] for j in range(k)]].T
            B = pds.lin_reg(*cols, target=[dev(Y[:, j]) for j in range(k)], add_bias=bias)
            for j in range(k):
                bo = orc.pl_lr(X, Y[:, j], add_bias=bias)
                if bo is None or B[f"target_{j}"] is None:   # the default gate (relative determinant 1e-12) refuses correlated wide frames
                    assert bo is None and B[f"target_{j}"] is None, (p, n, bias, k)
                    note("multi-null", 0.0)
                    continue
                note("multi", nrel(B[f"target_{j}"], bo))
            assert worst.get("multi", 0.0) < 1e-9, (p, n, bias, k)
            continue
        method = rng.choice(["ols", "ridge", "lasso", "enet", "nnls", "gated"])
        kw = {"ols": {}, "ridge": {"l2_reg": 0.3}, "lasso": {"l1_reg": 0.02, "tol": 1e-9, "max_iter": 5000},
              "enet": {"l1_reg": 0.02, "l2_reg": 0.05, "tol": 1e-9, "max_iter": 5000}, "nnls": {"positive": True, "tol": 1e-10, "max_iter": 5000},
              "gated": {"singular_x_tol": 1e-10}}[method]
        b = pds.lin_reg(*cols, target=dev(y), add_bias=bias, **kw)
        bo = orc.pl_lr(X, y, add_bias=bias, **kw)
        if bo is None or b is None:   # the gate: both sides must refuse
            assert bo is None and b is None, (method, p, n, bias)
            note("lr/gated-null", 0.0)
            continue
        e = nrel(b, bo)
        note("lr/" + method, e)
        assert e < (1e-7 if method in ("lasso", "enet", "nnls") else 1e-9), (method, p, n, bias, e)
    elif kind == "f32":
        # f32 frames (LIN_REG_EXPR_F64 = False): single fits and Gram records at every width class, against the f64 truth of the f32 frame
        p = int(rng.choice([3, 8, 16, 17, 24, 32, 33, 48, 64, 80]))
        n = int(rng.integers(max(6 * p, 100), 120_000))
        bias = bool(rng.integers(0, 2))
        X = rng.normal(size=(n, p)).astype(np.float32)
        y = (X.astype(np.float64) @ rng.normal(size=p) + 0.3 + 0.1 * rng.normal(size=n)).astype(np.float32)
        pds.config.LIN_REG_EXPR_F64 = False
        try:
            M = pds.gram_moments(*[dev(X[:, j]) for j in range(p)], target=dev(y))
            b = pds.lin_reg(*[dev(X[:, j]) for j in range(p)], target=dev(y), add_bias=bias, singular_x_tol=0.0)
        finally:
            pds.config.LIN_REG_EXPR_F64 = True
        Z = np.c_[X.astype(np.float64), np.ones(n), y.astype(np.float64)]
        note("f32/gram", nrel(M, Z.T @ Z))
        note("f32/lin_reg", nrel(b, orc.pl_lr(X.astype(np.float64), y.astype(np.float64), add_bias=bias, singular_x_tol=0.0)))
        # (Gram bar: the bf16 three-plane split of frames beyond 64 features is 2e-6 from the f64 Gram by construction -- DESIGN 4.6, the
        #  parity tests hold it to 3e-6 --; up to 64 features the f32 matrix instructions stay below 1e-6.  Contract: 1e-4 on coefficients.)
        bar = 3e-6 if p > 64 else 1e-6
        assert nrel(M, Z.T @ Z) < bar and worst["f32/lin_reg"] < 1e-4, (p, n, bias, nrel(M, Z.T @ Z), worst["f32/lin_reg"])
    elif kind == "grouped":
        # contiguous groups, 1 .. 64 features (17 .. 64: one wave per system in registers + the pivoted-QR pass over what it marks)
        p = int(rng.choice([2, 9, 16, 17, 20, 31, 32, 33, 47, 48, 63, 64]))
        bias = bool(rng.integers(0, 2))
        pp = p + bias
        G = int(rng.integers(1, 400))
        sizes = rng.integers(max(1, pp - 3), int(rng.choice([2, 6, 12])) * pp + 5, size=G)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        N = int(off[-1])
        X = rng.normal(size=(N, p)) + rng.normal(size=p) * float(rng.choice([0.0, 0.5]))
        y = X @ rng.normal(size=p) + 0.3 + 0.1 * rng.normal(size=N)
        l2 = float(rng.choice([0.0, 0.0, 0.2]))
        co, nu = pds.lin_reg_by(*[dev(X[:, j]) for j in range(p)], target=dev(y), group_offsets=off, add_bias=bias, l2_reg=l2)
        co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
        co_o, nu_o = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=bias, l2_reg=l2, nthreads=4)
        # (a group next to the 1e-12 gate may fall on either side of it in two correct factorisations: count those, bound them)
        mism = int((nu != nu_o).sum())
        note("grouped/gate-side-differs", mism / max(G, 1))
        assert mism <= max(1, G // 50), (p, bias, G, l2, mism)
        ok = ~nu & ~nu_o & (sizes >= 2 * pp + 8)
        if ok.any():
            e = float(np.max(np.linalg.norm(co[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)))
            note("grouped/p<=16" if p <= 16 else "grouped/17..64", e)
            assert e < 1e-7, (p, bias, G, l2, e)
    elif kind == "report":
        p = int(rng.choice([1, 3, 8, 16, 17, 20, 32, 33, 48, 64]))  # 17 .. 64: the fused one-stream robust errors (moments_mid.hip)
        n = int(rng.integers(20 * p + 50, 80_000))
        bias = bool(rng.integers(0, 2))
        se = str(rng.choice(["se", "hc0", "hc1", "hc2", "hc3"]))
        X = rng.normal(size=(n, p))
        y = X @ rng.normal(size=p) + (0.4 if bias else 0.0) + 0.3 * rng.normal(size=n) * (0.5 + np.abs(X[:, 0]))
        r = pds.lin_reg_report(*[dev(X[:, j]) for j in range(p)], target=dev(y), add_bias=bias, std_err=se)
        ro = orc.lin_reg_report(np.c_[X, np.ones(n)] if bias else X, y, y_var=float(np.var(y, ddof=1)), std_err=se)
        key = {"se": "std_err"}.get(se, f"{se}_se")
        e = max(nrel(r["beta"], ro["beta"]), nrel(r[key], ro["std_err"]))
        note("report/" + se, e)
        assert e < 1e-9, (se, p, n, bias, e)
    elif kind in ("rolling", "recursive"):
        p = int(rng.choice([1, 2, 4, 7, 8, 10, 12, 14, 30]))
        bias = bool(rng.integers(0, 2))
        pp = p + bias
        w = int(rng.integers(max(2 * pp + 5, 20), 400))
        n = int(rng.integers(max(4 * pp + 300, w + 50), 40_000))  # (a window longer than the frame is an argument error, not a case)
        X = rng.random((n, p))
        y = X @ rng.normal(size=p) + 0.2 + 0.01 * rng.normal(size=n)
        lam = float(rng.choice([0.0, 0.1]))
        Xb = np.c_[X, np.ones(n)] if bias else X
        cols = [dev(X[:, j]) for j in range(p)]
        if kind == "rolling":
            co, pr, va = pds.rolling_lin_reg(*cols, target=dev(y), window_size=w, add_bias=bias, l2_reg=lam)
            ref = orc.rolling_lr(Xb, y, w, l2_reg=lam)
        else:
            co, pr, va = pds.recursive_lin_reg(*cols, target=dev(y), start_with=w, add_bias=bias, l2_reg=lam)
            ref = orc.recursive_lr(Xb, y, w, l2_reg=lam)
        co = co.cpu().numpy()[w - 1:]
        e = float(np.max(np.linalg.norm(co - ref, axis=1) / np.linalg.norm(ref, axis=1)))
        note(kind, e)
        assert e < 1e-7, (kind, p, bias, n, w, lam, e)   # the reference's chain itself drifts ~1e-10 from the direct solves
    else:
        p = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 10, 12, 16]))
        G = int(rng.integers(2, 3000))
        sizes = rng.integers(0, int(rng.choice([5, 60, 400])), size=G) + (p + 2)
        keys = np.repeat(rng.permutation(G).astype(np.int64) * 7 - 1000, sizes)
        N = len(keys)
        perm = rng.permutation(N)
        X = rng.normal(size=(N, p))
        y = X @ rng.normal(size=p) + 0.5 + 0.1 * rng.normal(size=N)
        bias = bool(rng.integers(0, 2))
        k, co, nu = pds.lin_reg_by_key(*[dev(X[perm, j]) for j in range(p)], target=dev(y[perm]), key=dev(keys[perm]), add_bias=bias)
        k, co, nu = k.cpu().numpy(), co.cpu().numpy(), nu.cpu().numpy().astype(bool)
        uk = np.unique(keys)
        assert np.array_equal(k, uk)
        for g in rng.choice(len(uk), size=min(20, len(uk)), replace=False):
            m = keys == uk[g]
            if m.sum() >= 3 * (p + bias) + 5 and not nu[g]:
                note("keyed", nrel(co[g], orc.pl_lr(X[m], y[m], add_bias=bias)))
        assert worst.get("keyed", 0.0) < 1e-8, (p, G, bias)
        # per-row predictions in the frame's (shuffled) order: the call's own coefficients applied to every row
        pr, rs, rn = pds.lin_reg_by_key_pred(*[dev(X[perm, j]) for j in range(p)], target=dev(y[perm]), key=dev(keys[perm]), add_bias=bias)
        pr, rn = pr.cpu().numpy(), rn.cpu().numpy().astype(bool)
        gi = np.searchsorted(uk, keys[perm])
        assert np.array_equal(rn, nu[gi]), (p, G, bias, N)
        Xb = np.c_[X[perm], np.ones(N)] if bias else X[perm]
        live = ~rn
        if live.any():
            own = np.einsum("ij,ij->i", Xb[live], co[gi[live]])
            sc = np.linalg.norm(Xb[live], axis=1) * np.linalg.norm(co[gi[live]], axis=1) + 1e-300
            note("keyed-pred" + ("/partition" if N >= 65536 else "/sort"), float(np.max(np.abs(pr[live] - own) / sc)))
            assert worst["keyed-pred" + ("/partition" if N >= 65536 else "/sort")] < 1e-9, (p, G, bias, N)
print("configurations per entry point and the worst normwise distance to the oracle:")
for k in sorted(count):
    print(f"  {k:14s} {count[k]:5d}   {worst[k]:.2e}")
