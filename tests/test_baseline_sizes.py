"""
BASELINE.json's configs at their configured sizes, against the ORACLE (not only through size-independent properties): the
frames are SURVEY.md 8(d)'s synthetic inputs (tools/synth.py, seeded, generated on the GPU); the HIP path runs on the whole
frame, and a prefix / sample large enough to be meaningful and small enough for the CPU restatement to finish in seconds is
brought to the host and compared.  Tolerances are the contract's: 1e-10 (f64) normwise, 1e-4 (f32).
"""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

pytestmark = pytest.mark.gpu

F64_TOL = 1e-10
F32_TOL = 1e-4


@pytest.fixture(scope="module")
def pds():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import polars_ds_extension_amd as m

    m.config.LIN_REG_EXPR_F64 = True
    return m


def _threads(orc):
    return max(1, min(64, orc.max_threads()))


def _rowrel(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64), axis=1) / np.linalg.norm(np.asarray(b, np.float64), axis=1)


# ------------------------------------------------------------------------------------------ the headline config of bench.py
def test_headline_config_against_oracle(pds, orc):
    """
    BASELINE.json's metric config, exactly as bench.py times it: group_by(key).agg(lin_reg) on 1e6 groups x 100 rows x 16 f64
    features (1e8 rows, the frame of tools/synth.headline_frame with bench.py's seed).  The HIP path fits all 1e6 groups; 2e5
    of them -- the first 1e5, 5e4 from the middle, the last 5e4 -- are brought to the host and fitted by the oracle's
    per-group gated col-piv QR: null flags equal, every group's coefficients within 1e-10 normwise.
    """
    import torch

    import synth

    G, R, P = 1_000_000, 100, 16
    xs, y = synth.headline_frame(G, R, P, seed=1234)
    off = torch.arange(0, G * R + 1, R, dtype=torch.int64, device="cuda")
    co, nu = pds.lin_reg_by(*xs, target=y, group_offsets=off, add_bias=False)
    assert co.shape == (G, P) and int(nu.sum().item()) == 0
    worst, checked = 0.0, 0
    for g0, g1 in ((0, 100_000), (475_000, 525_000), (950_000, 1_000_000)):
        r0, r1 = g0 * R, g1 * R
        host = [y[r0:r1].cpu().numpy()] + [x[r0:r1].cpu().numpy() for x in xs]
        off_h = np.arange(0, (g1 - g0) * R + 1, R, dtype=np.int64)
        co_o, nu_o = orc.grouped_lr(host, off_h, add_bias=False, tol=1e-12, nthreads=_threads(orc))
        assert np.array_equal(nu[g0:g1].cpu().numpy().astype(bool), nu_o) and not nu_o.any()
        err = _rowrel(co[g0:g1].cpu().numpy(), co_o)
        worst = max(worst, float(err.max()))
        checked += g1 - g0
    print(f"headline 1e6 x 100 x 16: {checked} groups vs oracle, max normwise rel {worst:.2e}")
    assert checked == 200_000 and worst < F64_TOL
    # the same step from the KEY COLUMN (bench.py `grouped_by_key`): sorted int64 keys with gaps -- the one-pass order check + run marks,
    # the scan and the mask pass must reproduce the 1e6 offsets exactly, so the fused kernel gives the same bits; and a frame whose LAST
    # two keys are swapped (an inversion only the final piece sees) must still come back with every key in order
    keys = (torch.arange(G, dtype=torch.int64, device="cuda") * 3 - 17).repeat_interleave(R)
    k1, co1, nu1 = pds.lin_reg_by_key(*xs, target=y, key=keys, max_groups=G)
    assert torch.equal(k1, torch.arange(G, dtype=torch.int64, device="cuda") * 3 - 17)
    assert torch.equal(co1, co) and torch.equal(nu1, nu)
    del k1, co1, nu1
    keys[-1], keys[-R - 1] = keys[-R - 1].clone(), keys[-1].clone()  # one row of the last group moves into the one before: not ordered
    k2, co2, nu2 = pds.lin_reg_by_key(*xs, target=y, key=keys, max_groups=G)
    assert torch.equal(k2, torch.arange(G, dtype=torch.int64, device="cuda") * 3 - 17) and int(nu2.sum().item()) == 0
    assert torch.equal(co2[: G - 2], co[: G - 2]) or float((co2[: G - 2] - co[: G - 2]).abs().max().item()) < 1e-9
    del keys, k2, co2, nu2
    # the same frame as ONE regression (configs[1]'s Gram build): moment matrix of a 1e7-row prefix against the oracle's
    # blocked Gram, and of the whole frame against an f64 torch reduction of sampled entries
    n_s = 10_000_000
    M = pds.gram_moments(*[x[:n_s] for x in xs], target=y[:n_s])
    Zs = [x[:n_s].cpu().numpy() for x in xs] + [np.ones(n_s), y[:n_s].cpu().numpy()]
    ref = orc.gram_cols(Zs, nthreads=_threads(orc))
    assert np.linalg.norm(M - ref) / np.linalg.norm(ref) < 1e-13
    Mf = pds.gram_moments(*xs, target=y)
    for i, j in ((0, 0), (3, 11), (15, 17), (7, 16), (17, 17)):
        zi = xs[i] if i < P else (torch.ones_like(y) if i == P else y)
        zj = xs[j] if j < P else (torch.ones_like(y) if j == P else y)
        want = float(torch.dot(zi, zj).item())
        assert abs(Mf[i, j] - want) <= 1e-11 * max(abs(want), float(torch.linalg.vector_norm(zi).item() * torch.linalg.vector_norm(zj).item()))


# ------------------------------------------------------------------------------------------ configs[2]: grouped, 8(d) C3 spec
def test_c3_spec_sorted_and_shuffled_keys_against_oracle(pds, orc):
    """
    1e6 groups, Poisson(100) sizes clipped to [16, 256], 8 features, 0.1 % collinear groups, int64 keys -- sorted and
    shuffled.  Null flags of ALL groups of the sample equal the oracle's; coefficients of every non-null group within 1e-10.
    """
    import torch

    import synth

    G, p, S = 1_000_000, 8, 200_000
    fr = synth.c3_frame(G, p, seed=2)
    xs, y, off = fr["xs"], fr["y"], fr["offsets"]
    assert 0.95e8 < fr["n_rows"] < 1.05e8 and int(fr["sizes"].min()) >= 16 and int(fr["sizes"].max()) <= 256
    co, nu = pds.lin_reg_by(*xs, target=y, group_offsets=off)
    # the gate fires exactly on the collinear groups (and nowhere else: every group has >= 16 rows for 8 features)
    assert bool((nu.bool() == fr["collinear"]).all()) and int(nu.sum().item()) == int(fr["collinear"].sum().item()) >= 900
    # ---- oracle on the first S groups (their rows are a prefix of the frame)
    n_s = int(off[S].item())
    host = [y[:n_s].cpu().numpy()] + [x[:n_s].cpu().numpy() for x in xs]
    off_h = off[: S + 1].cpu().numpy()
    co_o, nu_o = orc.grouped_lr(host, off_h, nthreads=_threads(orc))
    assert np.array_equal(nu[:S].cpu().numpy().astype(bool), nu_o)
    ok = ~nu_o
    err = _rowrel(co[:S].cpu().numpy()[ok], co_o[ok])
    print(f"C3 spec, sorted keys: {S} groups vs oracle, max normwise rel {err.max():.2e}, nulls {int(nu_o.sum())}")
    assert err.max() < F64_TOL
    assert bool(torch.isnan(co[nu.bool()]).all())
    # ---- the same frame with shuffled rows, grouped by the key column on the device
    perm = torch.randperm(fr["n_rows"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(22))
    keys_s = fr["keys"][perm] * 3 - 1_000_000  # (keys need not be 0..G-1)
    xs_s = [x[perm] for x in xs]
    y_s = y[perm]
    del perm
    k2, co2, nu2 = pds.lin_reg_by_key(*xs_s, target=y_s, key=keys_s)
    assert k2.shape[0] == G and bool((k2 == torch.arange(G, device="cuda") * 3 - 1_000_000).all())
    assert bool((nu2 == nu).all())
    okd = ~nu.bool()
    d = (co2[okd] - co[okd]).norm(dim=1) / co[okd].norm(dim=1)  # same groups, rows summed in another order
    print(f"C3 spec, shuffled keys vs sorted: max normwise rel {float(d.max()):.2e}")
    assert float(d.max()) < F64_TOL
    err2 = _rowrel(co2[:S].cpu().numpy()[ok], co_o[ok])
    assert err2.max() < F64_TOL


# ------------------------------------------------------------------------------------------ configs[3]: rolling, w = 256
@pytest.mark.parametrize("lam", [0.0, 0.1])
def test_c4_full_frame_prefix_against_the_reference_chain(pds, orc, lam):
    """1e8 rows x 8 features, window 256: the first 1e6 output rows against faer_rolling_lr's Woodbury chain (oracle)."""
    import synth

    n, p, w, m = 100_000_000, 8, 256, 1_000_000
    fr = synth.c4_frame(n, p, seed=3)
    co, pr, va = pds.rolling_lin_reg(*fr["xs"], target=fr["y"], window_size=w, l2_reg=lam)
    assert not bool(va[: w - 1].any()) and bool(va[w - 1:].all())
    Xh = np.stack([x[:m].cpu().numpy() for x in fr["xs"]], axis=1)
    yh = fr["y"][:m].cpu().numpy()
    ref = orc.rolling_lr(Xh, yh, w, l2_reg=lam)  # rows w-1 .. m-1
    got = co[w - 1: m].cpu().numpy()
    err = _rowrel(got, ref)
    print(f"C4 lambda={lam}: 1e6-row prefix vs the reference chain, max normwise rel {err.max():.2e}")
    assert err.max() < F64_TOL
    # pred_i = x_i . coeffs_i (linear_regression.rs:1241-1269): the contract's 1e-10 on the coefficients, propagated -- the
    # alternating-sign coefficients make |pred_i| itself as small as 1e-3 of |x_i| |coeffs_i|
    pred_ref = np.einsum("ij,ij->i", Xh[w - 1:], ref)
    scale = np.linalg.norm(Xh[w - 1:], axis=1) * np.linalg.norm(ref, axis=1)
    assert np.max(np.abs(pr[w - 1: m].cpu().numpy() - pred_ref) / scale) < F64_TOL
    # far end of the frame: direct window solves (the chain would have to run 1e8 steps on the host to get there)
    for i in (n - 1, n - 123_457, n // 2):
        A = np.stack([x[i - w + 1: i + 1].cpu().numpy() for x in fr["xs"]], axis=1)
        b = fr["y"][i - w + 1: i + 1].cpu().numpy()
        direct = np.linalg.solve(A.T @ A + lam * np.eye(p), A.T @ b)
        assert np.linalg.norm(co[i].cpu().numpy() - direct) / np.linalg.norm(direct) < 1e-9


# ------------------------------------------------------------------------------------------ configs[1]: single OLS + report
def test_c2_prefix_against_oracle(pds, orc):
    """C2 data (U(0,1) features, two zero coefficients): 1e7-row prefix, lin_reg and lin_reg_report(SE) against the oracle."""
    import synth

    n, p = 10_000_000, 16
    fr = synth.c2_frame(n, p, seed=1)
    Xh = np.stack([x.cpu().numpy() for x in fr["xs"]], axis=1)
    yh = fr["y"].cpu().numpy()
    b = pds.lin_reg(*fr["xs"], target=fr["y"], add_bias=True)
    bo = orc.pl_lr(Xh, yh, add_bias=True)
    assert np.linalg.norm(b - bo) / np.linalg.norm(bo) < F64_TOL
    r = pds.lin_reg_report(*fr["xs"], target=fr["y"], add_bias=True)
    ro = orc.lin_reg_report(np.c_[Xh, np.ones(n)], yh)
    assert np.linalg.norm(r["beta"] - ro["beta"]) / np.linalg.norm(ro["beta"]) < F64_TOL
    assert np.max(np.abs(r["std_err"] - ro["std_err"]) / ro["std_err"]) < F64_TOL
    # t_i = beta_i / se_i.  The contract on beta is normwise (an elementwise relative error on beta_3 = beta_11 = 0 is
    # ill-posed), so what it promises for t is |dt_i| <= 1e-10 (|beta| / se_i + |t_i|); p = 2 sf(|t|) moves by
    # 2 pdf(t) |dt| on top of the special functions, which are bit-identical to the oracle's (test_cabi_cpu).
    from scipy import stats

    dt_bound = F64_TOL * (np.linalg.norm(ro["beta"]) / ro["std_err"] + np.abs(ro["t"]))
    dt = np.abs(r["t"] - ro["t"])
    print(f"C2 1e7: max |dt| / bound {np.max(dt / dt_bound):.2e}; |t| in [{np.min(np.abs(ro['t'])):.2f}, {np.max(np.abs(ro['t'])):.0f}]")
    assert np.all(dt <= dt_bound)
    dof = n - (p + 1)
    dp_bound = 2.0 * stats.t.pdf(np.abs(ro["t"]), dof) * dt_bound + 1e-14 * ro["p"]
    assert np.all(np.abs(r["p>|t|"] - ro["p"]) <= dp_bound)
    assert np.sum((ro["p"] > 1e-3) & (ro["p"] < 0.999)) >= 2  # the two zero coefficients give non-trivial p-values
    # CI ends beta_i -+ t_crit se_i: 1e-10 (|beta| + t_crit se_i), the same propagation
    t_crit = float(orc.student_t_ppf(0.975, float(dof))) if dof < 1.4e7 else float(stats.t.ppf(0.975, dof))
    ci_bound = F64_TOL * (np.linalg.norm(ro["beta"]) + t_crit * ro["std_err"])
    assert np.all(np.abs(r["0.025"] - ro["ci_lo"]) <= ci_bound) and np.all(np.abs(r["0.975"] - ro["ci_hi"]) <= ci_bound)
    assert abs(np.ravel(r["r2"])[0] - ro["r2"]) < 1e-12
    # robust errors on the same prefix: HC1 (meat fused into the residual pass) and HC2 / HC3 (round 5: the leverages on the matrix
    # cores) against the oracle's per-row arithmetic -- and on the first 9 and 12 features, the widths whose matrix steps stop early
    for cols_n in (p, 12, 9):
        Xb = np.c_[Xh[:, :cols_n], np.ones(n)]
        for kind in ("hc1", "hc2", "hc3"):
            rk = pds.lin_reg_report(*fr["xs"][:cols_n], target=fr["y"], add_bias=True, std_err=kind)
            rok = orc.lin_reg_report(Xb, yh, std_err=kind)
            e_se = float(np.max(np.abs(rk[f"{kind}_se"] - rok["std_err"]) / rok["std_err"]))
            assert e_se < F64_TOL, (cols_n, kind, e_se)
            assert np.linalg.norm(rk["beta"] - rok["beta"]) / np.linalg.norm(rok["beta"]) < F64_TOL


# ------------------------------------------------------------------------------------------ configs[4]: elastic net, f32, p = 512
def test_c5_one_million_rows_against_oracle_f32_and_f64(pds, orc):
    """
    1e6 x 512 f32, AR(0.5) columns, l1 = l2 = 0.01, tol = 1e-5: against the oracle's all-f32 path (what "matches the
    reference f32 path" means) and against the f64 truth -- both distances printed.
    """
    import synth

    n, p = 1_000_000, 512
    fr = synth.c5_frame(n, p, seed=4)
    X, y = fr["X"], fr["y"]
    import os

    # two arithmetics for the f32 Gram beyond 16 features (moments_wide.hip): the default runs the products on the bf16 matrix
    # cores as exact three-plane splits (1.3x faster, Gram error ~3 ulp of f32), the context option "wide_f32_native" keeps
    # v_mfma_f32_32x32x2_f32 (the fmaf chain, ~1 ulp).  Both are held to the contract; the native one also to the
    # reference's own f32 distance.
    fits = {}
    pds.config.LIN_REG_EXPR_F64 = False
    try:
        for native in ("0", "1"):
            pds.default_context().set_option("wide_f32_native", int(native))
            fits[native] = pds.lin_reg(*[X[j] for j in range(p)], target=y, l1_reg=0.01, l2_reg=0.01, tol=1e-5)
    finally:
        pds.config.LIN_REG_EXPR_F64 = True
        pds.default_context().set_option("wide_f32_native", 0)
    b = fits["0"]
    assert b.dtype == np.float32
    Xh = np.asfortranarray(X.cpu().numpy().T)
    yh = y.cpu().numpy()
    nt = _threads(orc)
    o32 = orc.coordinate_descent(Xh, yh, 0.01, 0.01, False, 1e-5, 2000, False, nthreads=nt)
    truth = orc.coordinate_descent(Xh.astype(np.float64), yh.astype(np.float64), 0.01, 0.01, False, 1e-9, 2000, False, nthreads=nt)
    nrm = np.linalg.norm(truth)
    d_orc = np.linalg.norm(o32 - truth) / nrm
    for native, bb in fits.items():
        d_gpu, d_go = np.linalg.norm(bb - truth) / nrm, np.linalg.norm(bb - o32) / nrm
        print(f"C5 1e6 x 512 ({'f32 matrix cores' if native == '1' else 'bf16 x 3 split'}): gpu-truth {d_gpu:.2e}  "
              f"oracle_f32-truth {d_orc:.2e}  gpu-oracle_f32 {d_go:.2e}")
        assert d_gpu < F32_TOL
        # the native arithmetic: never further from the truth than the reference's own f32 arithmetic; the split: within
        # 1e-5 (measured 4.3e-6), a tenth of the contract
        assert d_gpu <= (max(d_orc, 2e-6) if native == "1" else 1e-5)
        assert np.array_equal(np.abs(bb) > 1e-6, np.abs(truth) > 1e-6)  # same support
    assert np.array_equal(np.abs(b) > 1e-6, np.abs(truth) > 1e-6)  # same support


@pytest.mark.parametrize("n", [10_000_000, 30_000_000])
def test_c5_configured_size_gram_and_descent(pds, orc, n):
    """
    configs[4] at its configured size, 1e7 rows x 512 f32 features (20.5 GB), and at three times that (61 GB: the verdict's "largest
    f32 frame" leg -- the split arithmetic's Gram distance does NOT grow with the frame: every 8192-row split is summed in f32 on
    the matrix core and the splits in f64, so the distance is that of one split whatever their number; tools/wide_split_growth.py,
    profiles/r04_wide_split_growth.txt: 1.96e-6 at 1e6, 1e7 and 3e7 rows of this frame).  A CPU fit of the whole frame is minutes, so the
    two stages are held separately, each against an independent reference, for BOTH f32 Gram arithmetics:
      Gram      the library's f32 moment matrix against an f64 Gram of the same f32 data formed by torch (hipBLAS dgemm over
                1e6-row chunks): whole matrix, Frobenius; plus the column-sum / X'y / y'y borders;
      descent   the library's coefficients against the ORACLE's coordinate descent (orc_cd_from_gram, the restatement of
                lr_solvers.rs:426-538) run on the library's own Gram matrix -- the sweep order, soft threshold and stopping rule
                at p = 512 -- and against the f64 truth (oracle descent on the f64 Gram, tol 1e-9): 1e-4, same support.
    """
    import os

    import torch

    import synth

    p = 512
    fr = synth.c5_frame(n, p, seed=4)
    X, y = fr["X"], fr["y"]
    # ---- f64 reference moments of the f32 data
    G64 = torch.zeros(p, p, dtype=torch.float64, device="cuda")
    c64 = torch.zeros(p, dtype=torch.float64, device="cuda")
    cs64 = torch.zeros(p, dtype=torch.float64, device="cuda")
    for r0 in range(0, n, 1_000_000):
        Xc = X[:, r0: r0 + 1_000_000].double()
        yc = y[r0: r0 + 1_000_000].double()
        G64.addmm_(Xc, Xc.T)
        c64.add_(Xc @ yc)
        cs64.add_(Xc.sum(dim=1))
        del Xc, yc
    ysum = float(y.double().sum().item())
    yy = float((y.double() ** 2).sum().item())
    G64h, c64h, cs64h = G64.cpu().numpy(), c64.cpu().numpy(), cs64.cpu().numpy()
    truth, _, conv = orc.cd_from_gram(G64h, c64h, cs64h, ysum, float(n), 0.01, 0.01, False, 1e-9, 5000)
    assert conv
    nrm = np.linalg.norm(truth)
    cols = [X[j] for j in range(p)]
    pds.config.LIN_REG_EXPR_F64 = False
    try:
        for native in ("0", "1"):
            pds.default_context().set_option("wide_f32_native", int(native))
            name = "f32 matrix cores" if native == "1" else "bf16 x 3 split"
            M = np.asarray(pds.gram_moments(*cols, target=y), dtype=np.float64)  # Z'Z, Z = [X | 1 | y]
            assert M.shape == (p + 2, p + 2)
            d_g = np.linalg.norm(M[:p, :p] - G64h) / np.linalg.norm(G64h)
            d_c = np.linalg.norm(M[:p, p + 1] - c64h) / np.linalg.norm(c64h)
            d_s = np.linalg.norm(M[:p, p] - cs64h) / max(np.linalg.norm(cs64h), np.sqrt(n * p))  # (sums of N(0,1) columns: O(sqrt n))
            print(f"C5 {n:.0e} x 512 Gram ({name}): X'X {d_g:.2e}  X'y {d_c:.2e}  col sums {d_s:.2e}")
            # measured: 2.0e-6 for the split at EVERY frame length (a negative bias of ~2.6e-6 on the diagonal: the bf16 instruction's
            # f32 accumulation over a split's 512 chained steps; the three dropped plane products are below 2^-24 |x||y| and do not
            # show) and ~2.5e-7 for the f32 instructions.  What the contract binds is the coefficients (1e-4, below); the Gram bound
            # here is a regression guard at 2.5x the measured figure, the same for both frame lengths.
            assert d_g < (5e-6 if native == "0" else 5e-7) and d_c < 2e-6 and d_s < 1e-4
            assert abs(M[p, p] - n) < 0.5 and abs(M[p + 1, p + 1] - yy) / yy < 1e-6 and abs(M[p, p + 1] - ysum) <= 1e-6 * np.sqrt(n * yy / n)
            b = pds.lin_reg(*cols, target=y, l1_reg=0.01, l2_reg=0.01, tol=1e-5)
            assert b.dtype == np.float32 and b.shape == (p,)
            # the oracle's descent on the library's own Gram matrix (f64 arithmetic on f32-rounded moments; 2000 sweeps = the f32 cap)
            bo, _, _ = orc.cd_from_gram(M[:p, :p].copy(), M[:p, p + 1].copy(), M[:p, p].copy(), float(M[p, p + 1]), float(n), 0.01, 0.01,
                                        False, 1e-5, 2000)
            d_cd = np.linalg.norm(b - bo) / np.linalg.norm(bo)
            d_tr = np.linalg.norm(b - truth) / nrm
            print(f"C5 {n:.0e} x 512 descent ({name}): gpu - oracle CD on the same Gram {d_cd:.2e}; gpu - f64 truth {d_tr:.2e}; "
                  f"non-zeros {int((np.abs(b) > 1e-6).sum())}")
            assert d_cd < F32_TOL and d_tr < F32_TOL
            assert np.array_equal(np.abs(b) > 1e-6, np.abs(truth) > 1e-6)
    finally:
        pds.config.LIN_REG_EXPR_F64 = True
        pds.default_context().set_option("wide_f32_native", 0)


def test_ordered_keys_beyond_2_31_rows(pds):
    """`lin_reg_by_key` on an ORDERED int64 key column of 2^31 + 4096 rows x 1 feature (51.5 GB resident: keys, target, feature): the
    order check, the run marks and the fused fit index rows with 64 bits -- only the routes for unordered keys carry 32-bit row ranks
    (capi_grouped.hpp).  The reference's marshalling has no row bound (linear_regression.rs:151-267).  Checked: the group list, and the
    fits of groups at the front, across the 2^31 boundary and at the end against an f64 closed form on the same rows."""
    import torch

    free, _ = torch.cuda.mem_get_info()
    n = (1 << 31) + 4096
    if free < 64 * (1 << 30):
        pytest.skip("needs 64 GB of free HBM")
    dev_ = torch.device("cuda", 0)
    R = 4096
    G = n // R
    assert G * R == n
    gen = torch.Generator(device=dev_)
    gen.manual_seed(2031)
    x = torch.rand(n, dtype=torch.float64, device=dev_, generator=gen)
    y = torch.empty(n, dtype=torch.float64, device=dev_)
    step = 1 << 27
    for a in range(0, n, step):  # y = (1 + g mod 7) x + noise, group g = row // R; built in pieces (no 17 GB temporaries)
        b = min(n, a + step)
        g_of = torch.arange(a, b, device=dev_, dtype=torch.int64) // R
        y[a:b] = x[a:b] * (1.0 + (g_of % 7).double()) + 1e-3 * torch.randn(b - a, dtype=torch.float64, device=dev_, generator=gen)
        del g_of
    key = torch.empty(n, dtype=torch.int64, device=dev_)
    for a in range(0, n, step):
        b = min(n, a + step)
        key[a:b] = torch.arange(a, b, device=dev_, dtype=torch.int64) // R * 3 - 11  # ordered, sparse key values
    keys_out, co, nu = pds.lin_reg_by_key(x, target=y, key=key, max_groups=G + 8)
    torch.cuda.synchronize()
    assert keys_out.shape[0] == G and co.shape == (G, 1) and not bool(nu.any().item())
    assert int(keys_out[0].item()) == -11 and int(keys_out[-1].item()) == (G - 1) * 3 - 11
    assert bool((keys_out[1:] - keys_out[:-1] == 3).all().item())
    g_cut = (1 << 31) // R
    assert g_cut == G - 1  # (the last group lies entirely beyond row 2^31)
    for g in (0, 1, G // 2, g_cut - 2, g_cut - 1, g_cut):
        xs_, ys_ = x[g * R:(g + 1) * R], y[g * R:(g + 1) * R]
        ref = float((xs_ @ ys_) / (xs_ @ xs_))
        assert abs(float(co[g, 0].item()) - ref) <= 1e-12 * abs(ref), (g, float(co[g, 0].item()), ref)
        assert abs(ref - (1.0 + g % 7)) < 1e-3
