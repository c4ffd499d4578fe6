"""
Parity tests proper: the HIP path (through the C ABI, libpds_lstsq_hip.so) against the CPU oracle on the
same seeded inputs, against the committed golden vectors, and -- at BASELINE.json's full sizes -- through
size-independent properties.  Bar: 1e-10 relative (f64) / 1e-4 (f32), normwise on coefficient vectors
(elementwise relative error on a near-zero coefficient is ill-posed), elementwise with a
floor where stated.  All tests need a real MI355X.
"""
import numpy as np
import pytest

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


@pytest.fixture()
def f32(pds):
    pds.config.LIN_REG_EXPR_F64 = False
    yield
    pds.config.LIN_REG_EXPR_F64 = True


def dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def cols_of(X):
    return [dev(X[:, j]) for j in range(X.shape[1])]


def nrel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def frel(a, b, floor):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def hold_f32(name, gpu, o32, truth, d=None, slack=1.25):
    """The f32 contract (tests/test_f32_contract.py): within 1e-4 of the f64 truth, or -- where the reference's own all-f32
    arithmetic is further than that from the truth -- no further than the reference's f32 path is."""
    d = d or nrel
    dg, do = d(gpu, truth), d(o32, truth)
    print(f"{name}: gpu-truth {dg:.2e}  orc32-truth {do:.2e}  gpu-orc32 {d(gpu, o32):.2e}")
    assert dg <= F32_TOL or (do > F32_TOL and dg <= do * slack), f"{name}: gpu-truth {dg:.2e}, orc32-truth {do:.2e}"


def make_xy(rng, n, p, noise=0.01, zero_coefs=()):
    X = rng.random((n, p))
    beta = np.array([(-1.0) ** j * (0.05 + 0.03 * j) for j in range(p)])
    for j in zero_coefs:
        beta[j] = 0.0
    y = X @ beta + noise * rng.normal(size=n)
    return X, y, beta


# ------------------------------------------------------------------------------------------ Gram build
@pytest.mark.parametrize("n,p", [(1, 1), (5, 2), (127, 3), (128, 4), (129, 16), (1000, 7), (100_003, 16), (2_000_001, 11)])
def test_moments_f64(pds, orc, n, p):
    rng = np.random.default_rng(n * 31 + p)
    X, y, _ = make_xy(rng, n, p)
    M = pds.gram_moments(*cols_of(X), target=dev(y))
    ref = orc.gram(np.c_[X, np.ones(n), y])  # Z'Z with Z = [X | 1 | y]
    assert nrel(M, ref) < 1e-13
    assert np.allclose(M, M.T, rtol=0, atol=0) or nrel(M, M.T) < 1e-15
    w = rng.random(n) + 0.5
    Mw = pds.gram_moments(*cols_of(X), target=dev(y), weights=dev(w))
    Z = np.c_[X, np.ones(n), y]
    assert nrel(Mw, Z.T @ (Z * w[:, None])) < 1e-13


@pytest.mark.parametrize("n,p", [(5, 17), (63, 20), (64, 31), (1000, 32), (4097, 33), (100_003, 40), (65_536, 47), (70_001, 48),
                                 (300_017, 49), (200_000, 64)])
def test_moments_mid_width_f64(pds, orc, n, p):
    """17 .. 64 f64 features: the streaming multi-tile-column Gram kernel (moments_mid.hip) -- every tile-column count, ragged row
    counts incl. fewer rows than one half-tile, run-to-run reproducible; unaligned column starts (an Arrow slice)."""
    rng = np.random.default_rng(n + 7 * p)
    X, y, _ = make_xy(rng, n, p)
    Z = np.c_[X, np.ones(n), y]
    ref = Z.T @ Z
    M = pds.gram_moments(*cols_of(X), target=dev(y))
    assert nrel(M, ref) < 1e-13 and nrel(M, M.T) < 1e-15
    assert np.array_equal(M, pds.gram_moments(*cols_of(X), target=dev(y)))
    w = rng.random(n) + 0.5
    Mw = pds.gram_moments(*cols_of(X), target=dev(y), weights=dev(w))
    assert nrel(Mw, Z.T @ (Z * w[:, None])) < 1e-13
    if n > 2:  # columns that start 8 bytes into a 16-byte unit
        full = dev(np.ascontiguousarray(np.c_[X, y].T))  # (p + 1) x n
        M2 = pds.gram_moments(*[full[j, 1:] for j in range(p)], target=full[p, 1:])
        Z2 = Z[1:]
        assert nrel(M2, Z2.T @ Z2) < 1e-13


def test_moments_host_and_device_inputs_agree_bitwise(pds):
    rng = np.random.default_rng(3)
    X, y, _ = make_xy(rng, 300_017, 9)
    a = pds.gram_moments(*[np.ascontiguousarray(X[:, j]) for j in range(9)], target=y)
    b = pds.gram_moments(*cols_of(X), target=dev(y))
    c = pds.gram_moments(*cols_of(X), target=dev(y))
    assert np.array_equal(a, b) and np.array_equal(b, c)  # no atomics: run-to-run and space-to-space reproducible


def test_moments_f32(pds, orc, f32):
    rng = np.random.default_rng(5)
    X, y, _ = make_xy(rng, 250_001, 16)
    X32, y32 = X.astype(np.float32), y.astype(np.float32)
    M = pds.gram_moments(*cols_of(X32), target=dev(y32))
    Z = np.c_[X32.astype(np.float64), np.ones(len(y)), y32.astype(np.float64)]
    assert M.dtype == np.float32
    assert nrel(M, Z.T @ Z) < 2e-7  # f32 matrix-core tiles folded into f64 every 256 rows


@pytest.mark.parametrize("n,p", [(100_003, 17), (50_001, 32), (31, 40), (257, 48), (300_007, 64)])
def test_moments_mid_width_f32(pds, orc, f32, n, p):
    """17 .. 64 f32 features on the streaming kernel: the f32 values are widened on their way out of LDS (exact products, f64 sums on
    the f64 matrix instruction) and the record is rounded to f32 once -- one f32 rounding away from the f64 truth of the f32 frame;
    weights; ragged row counts; columns that start 4 bytes into a 16-byte unit."""
    rng = np.random.default_rng(n + 11 * p)
    X, y, _ = make_xy(rng, n, p)
    X32, y32 = X.astype(np.float32), y.astype(np.float32)
    Z = np.c_[X32.astype(np.float64), np.ones(n), y32.astype(np.float64)]
    M = pds.gram_moments(*cols_of(X32), target=dev(y32))
    assert M.dtype == np.float32 and nrel(M, Z.T @ Z) < 1.2e-7
    assert np.array_equal(M, pds.gram_moments(*cols_of(X32), target=dev(y32)))
    w32 = (rng.random(n) + 0.5).astype(np.float32)
    Mw = pds.gram_moments(*cols_of(X32), target=dev(y32), weights=dev(w32))
    assert nrel(Mw, Z.T @ (Z * w32.astype(np.float64)[:, None])) < 1.2e-7
    if n > 4:
        full = dev(np.ascontiguousarray(np.c_[X32, y32].T))
        M2 = pds.gram_moments(*[full[j, 1:] for j in range(p)], target=full[p, 1:])
        assert nrel(M2, Z[1:].T @ Z[1:]) < 1.2e-7
    # the fit on top of it (rank gate off: make_xy's columns are correlated enough for the f32 default gate): the f32 contract
    # against the f64 truth of the same f32 frame
    if n > 4 * p:
        b = pds.lin_reg(*cols_of(X32), target=dev(y32), add_bias=True, singular_x_tol=0.0)
        bt = orc.pl_lr(X32.astype(np.float64), y32.astype(np.float64), add_bias=True, singular_x_tol=0.0)
        assert nrel(b, bt) < F32_TOL


# ------------------------------------------------------------------------------------------ pl_lr dispatch
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize(
    "kw",
    [
        dict(),
        dict(l2_reg=0.1),
        dict(solver="svd"),
        dict(solver="choleskey"),
        dict(solver="cholesky"),  # the reference only knows the misspelling; this falls through to qr
        dict(l1_reg=0.01),
        dict(l1_reg=0.01, l2_reg=0.02),
        dict(positive=True),
        dict(l2_reg=0.3, positive=True),
        dict(l1_reg=0.005, positive=True),
        dict(singular_x_tol=0.0),
    ],
)
def test_lin_reg_methods(pds, orc, kw, bias):
    rng = np.random.default_rng(42)
    X, y, _ = make_xy(rng, 200_003, 8, noise=0.05)
    y = y + 0.5
    b = pds.lin_reg(*cols_of(X), target=dev(y), add_bias=bias, tol=1e-10, max_iter=5000, **kw)
    okw = dict(kw)
    okw.setdefault("singular_x_tol", 1e-12)
    bo = orc.pl_lr(X, y, add_bias=bias, tol=1e-10, max_iter=5000, **okw)
    assert b is not None and bo is not None
    assert nrel(b, bo) < F64_TOL
    assert frel(b, bo, 1e-3) < 1e-9


def test_lin_reg_weighted_and_pred(pds, orc):
    rng = np.random.default_rng(8)
    X, y, _ = make_xy(rng, 150_000, 6, noise=0.05)
    w = rng.random(len(y)) + 0.1
    b = pds.lin_reg(*cols_of(X), target=dev(y), add_bias=True, weights=dev(w))
    assert nrel(b, orc.pl_lr(X, y, add_bias=True, weights=w)) < F64_TOL
    pred, resid = pds.lin_reg(*cols_of(X), target=dev(y), add_bias=True, return_pred=True)
    bo = orc.pl_lr(X, y, add_bias=True)
    ref = np.c_[X, np.ones(len(y))] @ bo
    assert nrel(pred.cpu().numpy(), ref) < F64_TOL
    assert np.max(np.abs(resid.cpu().numpy() - (y - ref))) < 1e-11


def test_config1_reference_benchmark_shape(pds, orc):
    # BASELINE configs[0]: pds.lin_reg(x1..x4, target=y, add_bias=False) on a 100k-row random f64 frame
    rng = np.random.default_rng(208)
    X = rng.random((100_000, 4))
    y = X @ [0.5, 0.25, -0.15, 0.2] + 1e-4 * rng.random(100_000)
    b_host = pds.lin_reg(*[np.ascontiguousarray(X[:, j]) for j in range(4)], target=y)  # host buffers, like Arrow
    assert nrel(b_host, orc.pl_lr(X, y)) < F64_TOL
    assert np.allclose(b_host, [0.5, 0.25, -0.15, 0.2], atol=1e-4)


def test_gate(pds, orc):
    rng = np.random.default_rng(0)
    x1 = rng.normal(size=2000)
    X = np.c_[x1, 2.0 * x1, rng.normal(size=2000)]
    y = rng.normal(size=2000)
    for solver in ("qr", "svd", "choleskey"):
        assert pds.lin_reg(*cols_of(X), target=dev(y), solver=solver) is None
        assert orc.pl_lr(X, y, solver=solver) is None
    out = pds.lin_reg(*cols_of(X), target=dev(y), singular_x_tol=0.0)
    assert out is not None and out.shape == (3,)
    Xc = np.c_[np.zeros(2000), rng.normal(size=2000)]  # zero-variance column: non-positive diagonal
    assert pds.lin_reg(*cols_of(Xc), target=dev(y)) is None
    Xs = rng.normal(size=(5000, 7)) * 1e3  # large-scale features: the gate lives in log space
    ys = Xs @ rng.normal(size=7) + rng.normal(size=5000)
    b = pds.lin_reg(*cols_of(Xs), target=dev(ys), add_bias=True)
    assert nrel(b, orc.pl_lr(Xs, ys, add_bias=True)) < F64_TOL


def test_errors_match_reference_strings(pds):
    from polars_ds_extension_amd._lib import PdsError

    with pytest.raises(PdsError, match="#Data < #features"):
        pds.lin_reg(*[np.zeros(2)] * 3, target=np.zeros(2))
    with pytest.raises(PdsError, match="Empty data"):
        pds.lin_reg(np.zeros(0), target=np.zeros(0))
    with pytest.raises(ValueError, match="max_iter"):
        pds.lin_reg(np.ones(5), target=np.ones(5), max_iter=0)


def test_rcond(pds, orc):
    rng = np.random.default_rng(123)
    X = rng.normal(size=(5000, 3))
    y = X @ [0.3, -0.2, 1.1] + rng.normal(size=5000) * 0.1
    b, sv = pds.lin_reg_w_rcond(*cols_of(X), target=dev(y), rcond=0.3)
    bo, svo = orc.solve_lr_rcond(X, y, rcond=0.3)
    assert nrel(b, bo) < F64_TOL and nrel(sv, svo) < F64_TOL
    ref, _, _, s = np.linalg.lstsq(X, y, rcond=0.3)
    assert np.max(np.abs(b - ref)) < 1e-10 and nrel(sv, s) < 1e-10  # tests/test_linear_exprs.py:477-512


# ------------------------------------------------------------------------------------------ report
@pytest.mark.parametrize("p", list(range(1, 17)))
def test_report_hc_leverages_every_width(pds, orc, p):
    """HC2 / HC3 at every width of the small-frame kernel, with and without intercept, on a frame whose length is off the 128-row tile
    (ragged last tile) and below it: the leverages come from the matrix cores (moments.hip LEVM: the four 16 x 16 x 4 steps stop early for
    narrow f64 frames, the intercept is handled in closed form) -- robust errors against the oracle's per-row arithmetic, f64 and f32."""
    rng = np.random.default_rng(900 + p)
    for n in (100, 5_003, 70_001):
        if n <= 2 * p + 4:
            continue
        X = rng.normal(size=(n, p)) + 0.3
        y = X @ rng.normal(size=p) + 0.5 + (0.2 + np.abs(X[:, 0])) * rng.normal(size=n)  # heteroskedastic
        for bias in (False, True):
            Xb = np.c_[X, np.ones(n)] if bias else X
            for se in ("hc2", "hc3"):
                r = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=bias, std_err=se)
                ro = orc.lin_reg_report(Xb, y, std_err=se)
                assert nrel(r["beta"], ro["beta"]) < F64_TOL, (n, bias, se)
                assert frel(r[f"{se}_se"], ro["std_err"], 1e-12) < 1e-9, (n, bias, se, frel(r[f"{se}_se"], ro["std_err"], 1e-12))
    # f32 frames (the f32 matrix instruction's operand layout differs: every step carries four strided features)
    n = 20_011
    X = (rng.normal(size=(n, p)) + 0.3).astype(np.float32)
    y = (X.astype(np.float64) @ rng.normal(size=p) + 0.5 + (0.2 + np.abs(X[:, 0])) * rng.normal(size=n)).astype(np.float32)
    pds.config.LIN_REG_EXPR_F64 = False
    try:
        r32 = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=True, std_err="hc3")
    finally:
        pds.config.LIN_REG_EXPR_F64 = True
    ro = orc.lin_reg_report(np.c_[X.astype(np.float64), np.ones(n)], y.astype(np.float64), std_err="hc3")
    assert np.max(np.abs(np.asarray(r32["hc3_se"], np.float64) - ro["std_err"]) / ro["std_err"]) < F32_TOL


@pytest.mark.parametrize("se", ["se", "hc0", "hc1", "hc2", "hc3"])
@pytest.mark.parametrize("bias", [False, True])
def test_lin_reg_report(pds, orc, se, bias):
    rng = np.random.default_rng(2)
    n, p = 120_000, 8
    X, y, beta = make_xy(rng, n, p, noise=0.0, zero_coefs=(3, 6))  # two true zeros -> non-trivial p-values
    y = y + (0.4 if bias else 0.0) + 0.3 * rng.normal(size=n) * (0.5 + X[:, 0])  # heteroskedastic noise
    yv = float(np.var(y, ddof=1))
    r = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=bias, std_err=se, y_var=yv)
    Xb = np.c_[X, np.ones(n)] if bias else X
    ro = orc.lin_reg_report(Xb, y, y_var=yv, std_err=se)
    key = {"se": "std_err"}.get(se, f"{se}_se")
    assert r["features"][-1] == ("__bias__" if bias else "x8")
    assert nrel(r["beta"], ro["beta"]) < F64_TOL
    assert frel(r[key], ro["std_err"], 1e-12) < F64_TOL
    # t_i = beta_i / se_i.  The contract on beta is 1e-10 NORMWISE (an elementwise relative error on the two true-zero
    # coefficients is ill-posed) and 1e-10 elementwise on se, so what it promises for t is
    # |dt_i| <= 1e-10 (|beta| / se_i + |t_i|); p = 2 sf(|t|) moves by 2 pdf(t) |dt| (the special functions themselves are bit
    # identical, below); the CI ends beta_i -+ t_crit se_i move by 1e-10 (|beta| + t_crit se_i).  Same bounds as
    # tests/test_baseline_sizes.py::test_c2_prefix_against_oracle.
    from scipy import stats as _st

    beta_o, se_o, t_o = np.asarray(ro["beta"]), np.asarray(ro["std_err"]), np.asarray(ro["t"])
    dt_bound = F64_TOL * (np.linalg.norm(beta_o) / se_o + np.abs(t_o))
    assert np.all(np.abs(np.asarray(r["t"]) - t_o) <= dt_bound)
    dof_ = float(n - Xb.shape[1])
    dp_bound = 2.0 * _st.t.pdf(np.abs(t_o), dof_) * dt_bound + 1e-14 * np.asarray(ro["p"])
    assert np.all(np.abs(np.asarray(r["p>|t|"]) - np.asarray(ro["p"])) <= dp_bound)
    # ... and ALL of that slack is t's: the p-value the library reports is bit for bit the reference's special function of
    # the library's own t (stats.cpp restates beta.rs / gamma.rs operation by operation), so p meets the 1e-10 contract as a
    # function and differs from the oracle's p only through the 1e-12-level difference of the two t values
    dof = float(n - Xb.shape[1])
    p_of_own_t = np.array([2.0 * orc.student_t_sf(abs(float(tv)), dof) for tv in np.asarray(r["t"], np.float64)])
    assert np.array_equal(np.asarray(r["p>|t|"], np.float64), p_of_own_t)
    assert np.any((ro["p"] > 1e-6) & (ro["p"] < 0.999))
    t_crit = float(orc.student_t_ppf(0.975, dof))
    ci_bound = F64_TOL * (np.linalg.norm(beta_o) + t_crit * se_o)
    assert np.all(np.abs(np.asarray(r["0.025"]) - np.asarray(ro["ci_lo"])) <= ci_bound)
    assert np.all(np.abs(np.asarray(r["0.975"]) - np.asarray(ro["ci_hi"])) <= ci_bound)
    assert abs(r["r2"][0] - ro["r2"]) < 1e-12 and abs(r["adj_r2"][0] - ro["adj_r2"]) < 1e-12


@pytest.mark.parametrize("p", [1, 2, 3, 5, 8, 11, 16])
def test_report_and_pred_shapes(pds, orc, p):
    """Ragged row counts and every Gram-kernel packing (1 / 2 / 4 / 8 feature slots, 16): HC0 / HC1 form the residuals and the
    meat X' diag(e^2) X in ONE pass (moments_small_kernel WM = 2), SE / pred go through the streaming residual pass."""
    rng = np.random.default_rng(40 + p)
    for n in (130 + p, 128 * 37, 12_345 + p):
        X, y, _ = make_xy(rng, n, p, noise=0.0)
        y = y + 0.25 + 0.2 * rng.normal(size=n) * (0.5 + X[:, 0])
        for bias in (False, True):
            Xb = np.c_[X, np.ones(n)] if bias else X
            for se in ("se", "hc0", "hc1"):
                r = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=bias, std_err=se)
                ro = orc.lin_reg_report(Xb, y, std_err=se)
                key = {"se": "std_err"}.get(se, f"{se}_se")
                assert nrel(r["beta"], ro["beta"]) < F64_TOL, (n, bias, se)
                assert frel(r[key], ro["std_err"], 1e-12) < F64_TOL, (n, bias, se)
                assert abs(r["r2"][0] - ro["r2"]) < 1e-11
            pred, resid = pds.lin_reg(*cols_of(X), target=dev(y), add_bias=bias, return_pred=True)
            bo = orc.pl_lr(X, y, add_bias=bias)
            po = Xb @ bo
            assert nrel(pred.cpu().numpy(), po) < F64_TOL and nrel(resid.cpu().numpy(), y - po) < 1e-8


def test_rows_to_cols(pds):
    """pds_rows_to_cols_*: row-major matrix (row stride >= columns, host or HBM) -> contiguous columns in HBM, exact."""
    import ctypes as C

    import torch

    from polars_ds_extension_amd import _lib

    lib = _lib.load()
    ctx = pds.default_context()
    rng = np.random.default_rng(5)
    try:
        for chunk_mb in (256.0, 0.02):  # the second setting cuts host matrices into many row chunks
            lib.pds_set_host_staging(chunk_mb, 0.0)
            for n, p, ld in ((1, 1, 1), (63, 5, 5), (1000, 33, 40), (70_001, 17, 17), (257, 100, 100), (4096, 512, 512)):
                for dt, tdt, suf in ((np.float64, torch.float64, "_f64"), (np.float32, torch.float32, "_f32")):
                    A = rng.normal(size=(n, ld)).astype(dt)
                    fn = getattr(lib, "pds_rows_to_cols" + suf)
                    for space in (_lib.PDS_HOST, _lib.PDS_DEVICE):
                        src = A if space == _lib.PDS_HOST else torch.from_numpy(A).cuda()
                        ptr = src.ctypes.data if space == _lib.PDS_HOST else src.data_ptr()
                        stride = n + 3
                        out = torch.full((p, stride), -7.0, dtype=tdt, device="cuda")
                        ctx.follow_torch_stream(out.device)
                        _lib.check(fn(ctx._h, C.c_void_p(ptr), C.c_int64(ld), C.c_int64(n), C.c_int(p), C.c_int(space),
                                      C.c_void_p(out.data_ptr()), C.c_int64(stride)))
                        got = out.cpu().numpy()
                        assert np.array_equal(got[:, :n], A[:, :p].T), (n, p, ld, suf, space)
                        assert np.all(got[:, n:] == -7.0)
    finally:
        lib.pds_set_host_staging(256.0, 98304.0)


def test_host_frames_in_row_chunks(pds, orc):
    """PDS_HOST frames of more than one chunk go through a chunk-sized staging buffer (capi.hip, moments_from_host_chunked /
    pred_from_host_chunked): same answers as the whole-frame path, O(chunk) HBM.  The chunk is shrunk to 1024 rows here."""
    from polars_ds_extension_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(77)
    n, p = 30_001, 5
    X, y, _ = make_xy(rng, n, p, noise=0.05)
    y = y + 0.2
    w = rng.random(n) + 0.1
    hc = [np.ascontiguousarray(X[:, j]) for j in range(p)]
    whole = pds.lin_reg(*hc, target=y, add_bias=True)
    whole_m = pds.gram_moments(*hc, target=y)
    stage_before = pds.default_context()
    try:
        lib.pds_set_host_staging(0.05, 0.0)  # 30 chunks of 1024 rows
        b = pds.lin_reg(*hc, target=y, add_bias=True)
        assert nrel(b, orc.pl_lr(X, y, add_bias=True)) < F64_TOL and nrel(b, whole) < 1e-12
        assert np.array_equal(b, pds.lin_reg(*hc, target=y, add_bias=True))  # chunk records summed in a fixed order
        assert nrel(pds.gram_moments(*hc, target=y), whole_m) < 1e-13
        bw = pds.lin_reg(*hc, target=y, add_bias=True, weights=w)
        assert nrel(bw, orc.pl_lr(X, y, add_bias=True, weights=w)) < F64_TOL
        bl = pds.lin_reg(*hc, target=y, l1_reg=0.01, tol=1e-9, max_iter=2000)
        assert nrel(bl, orc.pl_lr(X, y, l1_reg=0.01, tol=1e-9, max_iter=2000)) < 1e-8
        # pred: resident (one PCIe trip) while the frame is under the limit, two chunked trips above it
        pr1, rs1 = pds.lin_reg(*hc, target=y, add_bias=True, return_pred=True)
        lib.pds_set_host_staging(0.05, 0.01)
        pr2, rs2 = pds.lin_reg(*hc, target=y, add_bias=True, return_pred=True)
        po = np.c_[X, np.ones(n)] @ orc.pl_lr(X, y, add_bias=True)
        for pr, rs in ((pr1, rs1), (pr2, rs2)):
            assert nrel(np.asarray(pr), po) < F64_TOL and nrel(np.asarray(rs), y - po) < 1e-8
        # f32 twin: chunk records stay in f64 until they are summed
        pds.config.LIN_REG_EXPR_F64 = False
        X32, y32 = X.astype(np.float32), y.astype(np.float32)
        b32 = pds.lin_reg(*[np.ascontiguousarray(X32[:, j]) for j in range(p)], target=y32, add_bias=True)
        assert b32.dtype == np.float32 and nrel(b32, orc.pl_lr(X32.astype(np.float64), y32.astype(np.float64), add_bias=True)) < F32_TOL
    finally:
        pds.config.LIN_REG_EXPR_F64 = True
        lib.pds_set_host_staging(256.0, 98304.0)
    assert stage_before is pds.default_context()


def test_wls_report(pds, orc):
    rng = np.random.default_rng(3)
    n = 90_000
    X, y, _ = make_xy(rng, n, 5, noise=0.2, zero_coefs=(2,))
    w = rng.random(n) + 0.1
    r = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=True, weights=dev(w), y_var=float(np.var(y, ddof=1)))
    ro = orc.wls_report(np.c_[X, np.ones(n)], y, w)
    assert nrel(r["beta"], ro["beta"]) < F64_TOL and frel(r["std_err"], ro["std_err"], 1e-12) < F64_TOL
    assert frel(r["p>|t|"], ro["p"], 1e-12) < 1e-8 and abs(r["r2"][0] - ro["r2"]) < 1e-12


def test_report_y_var_from_moments(pds):
    rng = np.random.default_rng(4)
    X, y, _ = make_xy(rng, 50_000, 4, noise=0.2)
    a = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=True)
    b = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=True, y_var=float(np.var(y, ddof=1)))
    assert abs(a["r2"][0] - b["r2"][0]) < 1e-10


def test_report_y_var_given_is_used_as_given_and_nan_propagates(pds):
    """A NaN var(y) handed over by the caller (a null `target.var()`, linear_regression.rs:836) is not silently replaced: r2 and
    adj_r2 come back NaN, everything that does not depend on it is unchanged.  Deriving it is an explicit request (y_var=None)."""
    rng = np.random.default_rng(41)
    X, y, _ = make_xy(rng, 20_000, 3, noise=0.2)
    a = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=True)
    b = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=True, y_var=float("nan"))
    assert np.isnan(b["r2"][0]) and np.isnan(b["adj_r2"][0]) and np.isfinite(a["r2"][0])
    assert np.array_equal(a["beta"], b["beta"]) and np.array_equal(a["std_err"], b["std_err"])


def test_report_f32_derived_variance_does_not_cancel(pds, f32):
    """|mean(y)| >> std(y): var(y) formed from f32-rounded sum y, sum y^2 would lose every digit (1e-7 * (mean / std)^2 = 10
    here); the library sums y in f64 for the f32 report, so r2 matches the one computed with numpy's centred variance."""
    rng = np.random.default_rng(42)
    n = 200_000
    X = rng.normal(size=(n, 3)).astype(np.float32)
    y = (1.0e4 + X @ np.array([0.5, -0.25, 0.125]) + 0.1 * rng.normal(size=n)).astype(np.float32)
    yv = float(np.var(y.astype(np.float64), ddof=1))
    a = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=True)
    b = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=True, y_var=yv)
    assert abs(float(a["r2"][0]) - float(b["r2"][0])) < 1e-5, (a["r2"], b["r2"])


def test_report_derived_variance_with_a_large_offset(pds):
    """f64, |mean(y)| / std(y) = 1e9 (timestamps, prices with an offset): the raw-moment formula (sum y^2 - (sum y)^2 / n) loses
    every digit there (1e-16 * 1e18); the derived var(y) comes from sums of y - y[0] instead and r2 matches the one computed with
    the centred variance -- what Polars' `target.var()` hands the reference."""
    rng = np.random.default_rng(43)
    n = 300_000
    X = rng.normal(size=(n, 3))
    y = 1.0e9 + X @ np.array([0.5, -0.25, 0.125]) + 0.1 * rng.normal(size=n)
    yv = float(np.var(y, ddof=1))
    a = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=True)
    b = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=True, y_var=yv)
    assert 0.5 < float(b["r2"][0]) < 1.0 and abs(float(a["r2"][0]) - float(b["r2"][0])) < 1e-6, (a["r2"][0], b["r2"][0])


def test_default_context_is_per_thread(pds):
    """Python threads calling the functional API concurrently get their own context (stream, workspace, staging)."""
    from concurrent.futures import ThreadPoolExecutor

    rng = np.random.default_rng(8)
    frames = [(rng.normal(size=(int(rng.integers(50, 5000)), 4)), rng.normal(size=4)) for _ in range(24)]

    def one(i):
        X, b = frames[i]
        y = X @ b + 0.25
        return pds.lin_reg(*[np.ascontiguousarray(X[:, j]) for j in range(4)], target=y, add_bias=True)

    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(one, list(range(24)) * 3))
    for i, r in enumerate(res):
        np.testing.assert_allclose(r, np.r_[frames[i % 24][1], 0.25], atol=1e-9)


# ------------------------------------------------------------------------------------------ grouped
@pytest.mark.parametrize("p,bias", [(1, False), (2, False), (3, True), (4, True), (5, False), (7, False), (7, True), (8, False),
                                    (8, True), (9, False), (15, True), (16, False), (16, True), (4, False)])
def test_grouped(pds, orc, p, bias):
    rng = np.random.default_rng(100 + p)
    G = 3000
    sizes = rng.integers(1, 300, size=G)
    sizes[::97] = rng.integers(0, p + 1, size=len(sizes[::97]))  # empty / too-small groups
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    X = rng.normal(size=(N, p))
    y = np.empty(N)
    for g in range(G):
        s = slice(off[g], off[g + 1])
        y[s] = X[s] @ rng.normal(size=p) + 0.1 * rng.normal(size=sizes[g]) + (0.7 if bias else 0.0)
    if p >= 2:
        for g in range(5, G, 211):  # collinear groups must come back null through the gate
            X[off[g] : off[g + 1], 1] = 2.0 * X[off[g] : off[g + 1], 0]
    co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias)
    co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
    co_o, nu_o = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=bias, nthreads=4)
    assert np.array_equal(nu, nu_o)
    assert nu.sum() > 10
    ok = ~nu
    # Every non-null group is held to 1e-10 normwise against the oracle's column-pivoted QR -- except where the group's own
    # conditioning makes two correct solvers differ by more than that: sizes go down to rows == features here, and a solve
    # of the normal equations is only determined to eps * cond(X'X).  Those groups (and only those) are held to that bound.
    err = np.linalg.norm(co[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)
    idx = np.flatnonzero(ok)
    loose = idx[err >= F64_TOL]
    for g in loose:
        sl = slice(off[g], off[g + 1])
        Xg = np.c_[X[sl], np.ones(sizes[g])] if bias else X[sl]
        bound = 64 * np.finfo(np.float64).eps * np.linalg.cond(Xg.T @ Xg)
        e = err[np.searchsorted(idx, g)]
        assert e < bound, f"group {g} ({sizes[g]} rows): err {e:.2e} above 64 eps cond(X'X) = {bound:.2e}"
    well = sizes[ok] >= 2 * (p + bias) + 8
    assert np.max(err[well]) < F64_TOL
    print(f"grouped p={p} bias={bias}: {ok.sum()} fitted groups, {len(loose)} judged by their conditioning bound, max err of the rest "
          f"{np.max(err[err < F64_TOL]):.2e}")
    assert np.isnan(co[nu]).all()


def _grouped_pred_oracle(orc, X, y, off, bias, weights=None, **kw):
    """pl_lr_pred per group (linear_regression.rs:704-820): pred = [X 1] beta, resid = y - pred; a null group is NaN / flagged"""
    n, G = len(y), len(off) - 1
    pred, resid, rn = np.full(n, np.nan), np.full(n, np.nan), np.ones(n, dtype=bool)
    pp = X.shape[1] + int(bias)
    for g in range(G):
        s = slice(int(off[g]), int(off[g + 1]))
        m = s.stop - s.start
        if m < pp or m == 0:
            continue
        b = orc.pl_lr(X[s], y[s], add_bias=bias, weights=None if weights is None else weights[s], **kw)
        if b is None:
            continue
        pr = X[s] @ b[: X.shape[1]] + (b[-1] if bias else 0.0)
        pred[s], resid[s], rn[s] = pr, y[s] - pr, False
    return pred, resid, rn


@pytest.mark.parametrize("p,bias", [(1, False), (2, True), (5, False), (8, True), (13, False), (16, True), (20, False)])
def test_grouped_pred(pds, orc, p, bias):
    """group_by(key).agg(lin_reg(..., return_pred=True)) for contiguous groups: ragged sizes incl. empty and too-small groups,
    collinear groups (all of their rows null), every compile-time feature count of the kernel and the run-time column loop."""
    rng = np.random.default_rng(300 + p)
    G = 1500
    sizes = rng.integers(1, 260, size=G)
    sizes[::89] = rng.integers(0, p + 1, size=len(sizes[::89]))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    X = rng.normal(size=(N, p))
    y = np.empty(N)
    for g in range(G):
        sl = slice(off[g], off[g + 1])
        y[sl] = X[sl] @ rng.normal(size=p) + 0.1 * rng.normal(size=sizes[g]) + (0.7 if bias else 0.0)
    if p >= 2:
        for g in range(5, G, 173):
            X[off[g]: off[g + 1], 1] = 2.0 * X[off[g]: off[g + 1], 0]
    pred, resid, rn, co, nu = pds.lin_reg_by_pred(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias)
    pred, resid, rn = pred.cpu().numpy(), resid.cpu().numpy(), rn.cpu().numpy().astype(bool)
    po, ro, rno = _grouped_pred_oracle(orc, X, y, off, bias)
    assert np.array_equal(rn, rno) and (p < 2 or rn.sum() > 10)
    assert np.isnan(pred[rn]).all() and np.isnan(resid[rn]).all()
    ok = ~rn
    # pred_i = x_i . beta: the contract's 1e-10 on beta (normwise), propagated -- |x_i| |beta| 1e-10; groups whose own conditioning
    # loosens beta (rows barely above features) are judged per group like test_grouped does
    co2, nu2 = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias)
    # (same kernels in both calls: the same bits.  17 .. 64 features: a group cut by a wave boundary is summed from partial records --
    #  two of them in either order are the same number, so this holds there too unless a group outgrows a whole wave's rows)
    assert np.array_equal(co.cpu().numpy()[~nu.cpu().numpy().astype(bool)], co2.cpu().numpy()[~nu2.cpu().numpy().astype(bool)])
    gid = np.repeat(np.arange(G), sizes)
    well = (sizes >= 2 * (p + bias) + 8)[gid] & ok
    Xb = np.c_[X, np.ones(N)] if bias else X
    bnorm = np.linalg.norm(np.nan_to_num(co.cpu().numpy()), axis=1)[gid]
    scale = np.linalg.norm(Xb, axis=1) * bnorm
    assert np.max(np.abs(pred[well] - po[well]) / scale[well]) < F64_TOL
    assert np.max(np.abs(resid[well] - ro[well]) / np.maximum(scale[well], np.abs(y[well]))) < F64_TOL
    # pred is exactly the library's own beta applied to the row (also in the loosely conditioned groups)
    own = np.einsum("ij,ij->i", Xb[ok], co.cpu().numpy()[gid[ok]])
    np.testing.assert_allclose(pred[ok], own, rtol=1e-12, atol=1e-12 * np.max(scale[ok]))
    # host-resident inputs give the same bits
    ph, rh, nh, _, _ = pds.lin_reg_by_pred(*[np.ascontiguousarray(X[:, j]) for j in range(p)], target=y, group_offsets=off, add_bias=bias)
    assert np.array_equal(ph[ok], pred[ok]) and np.array_equal(nh.astype(bool), rn)


def test_grouped_pred_by_key_shuffled_weighted_f32(pds, orc):
    """Keys in any row order: predictions land where the rows are; weights; the f32 twin."""
    rng = np.random.default_rng(91)
    G, p = 800, 6
    sizes = rng.integers(12, 150, size=G)
    key_of_group = rng.permutation(G) * 7 - 1000
    key = np.repeat(key_of_group, sizes).astype(np.int64)
    N = len(key)
    X = rng.normal(size=(N, p))
    y = X @ rng.normal(size=p) + 0.01 * key + 0.1 * rng.normal(size=N)
    w = rng.random(N) + 0.25
    perm = rng.permutation(N)
    kp, Xp, yp, wp = key[perm], X[perm], y[perm], w[perm]
    order = np.argsort(kp, kind="stable")
    off = np.concatenate([[0], np.cumsum(np.unique(kp, return_counts=True)[1])]).astype(np.int64)
    for weights in (None, wp):
        pred, resid, rn = pds.lin_reg_by_key_pred(*cols_of(Xp), target=dev(yp), key=dev(kp), add_bias=True,
                                                  weights=None if weights is None else dev(weights))
        po, ro, rno = _grouped_pred_oracle(orc, Xp[order], yp[order], off, True, weights=None if weights is None else weights[order])
        back = np.empty(N, dtype=np.int64)
        back[order] = np.arange(N)
        assert not rn.any().item() and not rno.any()
        scale = np.linalg.norm(np.c_[Xp, np.ones(N)], axis=1) * 3.0
        assert np.max(np.abs(pred.cpu().numpy() - po[back]) / scale) < F64_TOL
        assert np.max(np.abs(resid.cpu().numpy() - ro[back]) / scale) < F64_TOL
    # host frame, keys already ordered (no data movement) == shuffled call, row by row
    ph, rh, nh = pds.lin_reg_by_key_pred(*[np.ascontiguousarray(Xp[order][:, j]) for j in range(p)], target=yp[order], key=kp[order],
                                         add_bias=True, weights=wp[order])
    np.testing.assert_allclose(ph, pred.cpu().numpy()[order], rtol=1e-10, atol=1e-12)
    # f32
    pds.config.LIN_REG_EXPR_F64 = False
    try:
        p32, r32, n32 = pds.lin_reg_by_key_pred(*cols_of(Xp.astype(np.float32)), target=dev(yp.astype(np.float32)), key=dev(kp), add_bias=True)
    finally:
        pds.config.LIN_REG_EXPR_F64 = True
    po, _, _ = _grouped_pred_oracle(orc, Xp[order], yp[order], off, True)
    assert p32.dtype.is_floating_point and p32.element_size() == 4
    assert np.max(np.abs(p32.cpu().numpy().astype(np.float64) - po[back]) / scale) < F32_TOL


def test_context_options(pds):
    """pds_ctx_set_option (include/pds_lstsq.h): unknown names are refused; "keyed_sort" sends a shuffled frame through the stable sort +
    gather route -- two runs give the same BITS (the partition route's sums follow cursor atomics: equal to rounding only) -- and agrees
    with the default route to rounding."""
    import torch

    from polars_ds_extension_amd import _lib

    ctx = pds.Context(0)
    with pytest.raises(_lib.PdsError):
        ctx.set_option("no_such_option", 1)
    rng = np.random.default_rng(12)
    G, p = 2000, 5
    sizes = rng.integers(20, 120, size=G)
    key = np.repeat(np.arange(G, dtype=np.int64) * 2 + 3, sizes)
    N = len(key)
    assert N >= 1 << 16
    X = rng.normal(size=(N, p))
    y = X @ rng.normal(size=p) + 1e-3 * key + 0.1 * rng.normal(size=N)
    perm = rng.permutation(N)
    cols, yt, kt = cols_of(X[perm]), dev(y[perm]), dev(key[perm])
    k0, c0, n0 = pds.lin_reg_by_key(*cols, target=yt, key=kt, ctx=ctx)           # partition route
    ctx.set_option("keyed_sort", 1)
    k1, c1, n1 = pds.lin_reg_by_key(*cols, target=yt, key=kt, ctx=ctx)
    k2, c2, n2 = pds.lin_reg_by_key(*cols, target=yt, key=kt, ctx=ctx)
    ctx.set_option("keyed_sort", 0)
    assert torch.equal(c1, c2) and torch.equal(n1, n2) and torch.equal(k1, k2)
    assert torch.equal(k0, k1) and torch.equal(n0, n1)
    assert float((c0 - c1).abs().max().item()) < 1e-9
    ctx.close()


def test_partition_route_does_not_reserve_the_sort_routes_workspace(pds):
    """The workspace a shuffled dense-key fit reserves (`pds_ctx_workspace_bytes`, capi_grouped.hpp `need`): the partition route holds
    its record buffers -- not ALSO the sorted keys, ranks and gathered copy of the frame the sort route needs (a round-5 regression:
    a dangling else tied that reservation to the per-row prediction table).  The same frame through "keyed_sort" reserves more."""
    rng = np.random.default_rng(77)
    G, p = 4000, 8
    key = np.repeat(np.arange(G, dtype=np.int64), 64)
    N = len(key)
    X = rng.normal(size=(N, p))
    y = X @ rng.normal(size=p) + 0.1 * rng.normal(size=N)
    perm = rng.permutation(N)
    cols, yt, kt = cols_of(X[perm]), dev(y[perm]), dev(key[perm])
    frame_bytes = N * (p + 1) * 8
    ctx = pds.Context(0)
    pds.lin_reg_by_key(*cols, target=yt, key=kt, ctx=ctx, max_groups=G)
    part = ctx.workspace_bytes("keyed")
    ctx.close()
    ctx = pds.Context(0)
    ctx.set_option("keyed_sort", 1)
    pds.lin_reg_by_key(*cols, target=yt, key=kt, ctx=ctx, max_groups=G)
    srt = ctx.workspace_bytes("keyed")
    ctx.close()
    # sort route: 2 key copies + 2 rank arrays + a gathered copy of the frame twice over (columns + packed rows) > 2 frames;
    # partition route: one record per row (frame + ids) + bucket tables
    assert srt > 2 * frame_bytes
    assert 0 < part < 2 * frame_bytes and part < srt - frame_bytes, (part, srt, frame_bytes)


@pytest.mark.parametrize("p,bias,f32", [(1, False, False), (4, True, False), (8, False, False), (8, True, True), (11, True, False), (16, False, False)])
def test_by_key_pred_partition_route(pds, orc, p, bias, f32):
    """Shuffled rows, dense integer keys, >= 2^17 rows: per-row predictions of the PARTITION route (grouped_pred.hip MODE 3, round 5: the
    frame read where it lies, the row's dense id indexing an id-indexed copy of the coefficient block) -- every row's prediction lands
    where the row is; collinear and too-small groups give null rows (NaN + flag); sparse key values; both precisions; against the oracle's per-group
    fits and against the call on the same frame in key order (the fused kernel + grouped_pred's offsets form)."""
    rng = np.random.default_rng(4100 + 10 * p + bias)
    G = 3000
    sizes = rng.integers(p + bias + 6, 110, size=G)
    sizes[::97] = rng.integers(1, p + bias, size=len(sizes[::97])) if p + bias > 1 else 1  # fewer rows than coefficients -> null (p' > 1)
    key_of_group = (rng.permutation(G) * 3 - 500).astype(np.int64)
    key = np.repeat(key_of_group, sizes)
    N = len(key)
    assert N >= 1 << 17
    X = rng.normal(size=(N, p))
    y = X @ rng.normal(size=p) + 1e-3 * key + 0.1 * rng.normal(size=N) + (0.4 if bias else 0.0)
    off0 = np.concatenate([[0], np.cumsum(sizes)])
    if p >= 2:
        for g in range(11, G, 211):
            X[off0[g]: off0[g + 1], 1] = 2.0 * X[off0[g]: off0[g + 1], 0]
    perm = rng.permutation(N)
    kp, Xp, yp = key[perm], X[perm], y[perm]
    dt = np.float32 if f32 else np.float64
    pds.config.LIN_REG_EXPR_F64 = not f32
    try:
        pred, resid, rn = pds.lin_reg_by_key_pred(*cols_of(Xp.astype(dt)), target=dev(yp.astype(dt)), key=dev(kp), add_bias=bias)
        order = np.argsort(kp, kind="stable")
        ps, rs, ns = pds.lin_reg_by_key_pred(*cols_of(Xp[order].astype(dt)), target=dev(yp[order].astype(dt)), key=dev(kp[order]), add_bias=bias)
    finally:
        pds.config.LIN_REG_EXPR_F64 = True
    pred, resid, rn = pred.cpu().numpy().astype(np.float64), resid.cpu().numpy().astype(np.float64), rn.cpu().numpy().astype(bool)
    ps, ns = ps.cpu().numpy().astype(np.float64), ns.cpu().numpy().astype(bool)
    uk, cnt = np.unique(kp, return_counts=True)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    po, ro, rno = _grouped_pred_oracle(orc, Xp[order], yp[order], off, bias)
    back = np.empty(N, dtype=np.int64)
    back[order] = np.arange(N)
    assert np.array_equal(rn, rno[back]) and (p + bias == 1 or rn.sum() > 20)
    assert np.isnan(pred[rn]).all() and np.isnan(resid[rn]).all()
    # the same null rows and -- to the two routes' different summation orders -- the same predictions as the ordered call
    assert np.array_equal(ns, rno)
    gid = np.repeat(np.arange(len(uk)), cnt)
    well = ((cnt >= 2 * (p + bias) + 8)[gid] & ~rno)[back]
    scale = (np.linalg.norm(np.c_[Xp, np.ones(N)], axis=1) * 3.0)
    tol = F32_TOL if f32 else F64_TOL
    assert np.max(np.abs(pred[well] - po[back][well]) / scale[well]) < tol
    assert np.max(np.abs(resid[well] - ro[back][well]) / np.maximum(scale[well], np.abs(yp[well]))) < tol
    assert np.max(np.abs(pred[well] - ps[back][well]) / scale[well]) < tol
    # resid = y - pred on every fitted row
    okr = ~rn
    np.testing.assert_allclose(resid[okr], yp.astype(dt).astype(np.float64)[okr] - pred[okr], rtol=0, atol=(1e-5 if f32 else 1e-12) * (1 + np.abs(yp[okr]).max()))


@pytest.mark.parametrize("n_ctx", [2, 3])
def test_exchange_steps_compose_the_multi_device_paths_through_the_c_abi(pds, orc, n_ctx):
    """pds_allreduce_sum_* / pds_scatter_rows_* / pds_gather_* (include/pds_lstsq.h, round 4): the three exchange steps of SURVEY.md
    8(e) for a host that drives several devices from one process and has no collective library -- here N contexts on device 0.
    (1) C2: a frame on rank 0 scattered by row ranges, every rank's moment block, all-reduce, the replicated solve = the whole-frame
    fit (oracle, 1e-10); (2) C4: prefix mode = the exclusive scan that seeds the row-sharded expanding fit; (3) C3: the ranks'
    coefficient blocks gathered in rank order."""
    import torch

    rng = np.random.default_rng(900 + n_ctx)
    n, p = 600_000, 6
    X = rng.normal(size=(n, p))
    y = X @ rng.normal(size=p) + 0.7 + 0.1 * rng.normal(size=n)
    ctxs = [pds.Context(0) for _ in range(n_ctx)]
    cols_dev = [dev(y)] + cols_of(X)
    bounds = [0] + [int(n * (c + 1) / n_ctx) + (7 if c + 1 < n_ctx else 0) for c in range(n_ctx)]
    shards = pds.scatter_rows(ctxs, cols_dev, bounds)
    for c in range(n_ctx):
        assert all(torch.equal(s, full[bounds[c]:bounds[c + 1]]) for s, full in zip(shards[c], cols_dev))
    moms = [pds.gram_moments(*sh[1:], target=sh[0], ctx=ctxs[c], out_device=True).contiguous() for c, sh in enumerate(shards)]
    parts = [m.clone() for m in moms]
    pds.allreduce_sum(ctxs, moms)
    want = sum(parts)
    for m in moms:
        assert torch.allclose(m, want, rtol=1e-14, atol=0.0)
    b = pds.lin_reg_from_moments(moms[n_ctx - 1], add_bias=True, ctx=ctxs[n_ctx - 1])
    b = b.cpu().numpy() if hasattr(b, "cpu") else np.asarray(b)
    assert nrel(b.ravel(), orc.pl_lr(X, y, add_bias=True)) < F64_TOL
    # prefix mode: rank c receives the sum of the ranks in front of it (rank 0: zeros)
    pre = [m.clone() for m in parts]
    pds.allreduce_sum(ctxs, pre, prefix=True)
    run = torch.zeros_like(parts[0])
    for c in range(n_ctx):
        assert torch.allclose(pre[c], run, rtol=1e-14, atol=0.0)
        run = run + parts[c]
    # the seeded expanding fit of the last shard = the tail of the whole-frame expanding fit
    co_full, pr_full, va_full = pds.recursive_lin_reg(*cols_of(X), target=dev(y), start_with=20, add_bias=True)
    c = n_ctx - 1
    co_s, pr_s, va_s = pds.recursive_lin_reg(*shards[c][1:], target=shards[c][0], start_with=20, add_bias=True, seed_moments=pre[c], ctx=ctxs[c])
    assert np.max(np.abs(co_s.cpu().numpy() - co_full.cpu().numpy()[bounds[c]:])) < 1e-9
    # gather: blocks of different lengths, rank order
    blocks = [torch.arange(100 * (c + 1), dtype=torch.float64, device="cuda") + 1000.0 * c for c in range(n_ctx)]
    got = pds.gather(ctxs, blocks)
    assert torch.equal(got, torch.cat(blocks))
    f32b = [torch.full((5,), float(c + 1), dtype=torch.float32, device="cuda") for c in range(n_ctx)]
    pds.allreduce_sum(ctxs, f32b)
    assert all(torch.equal(t, torch.full((5,), float(n_ctx * (n_ctx + 1) // 2), dtype=torch.float32, device="cuda")) for t in f32b)
    for cx in ctxs:
        cx.close()


def test_exchange_steps_and_unordered_by_key_across_two_devices(pds, orc):
    """The CROSS-DEVICE legs (hipMemcpyPeerAsync in pds_allreduce_sum_* / pds_scatter_rows_* / pds_gather_* and the peer table copy of the
    unordered multi-context by-key route, capi_multi.hpp): contexts on two different devices of one process.  Skipped on a one-GPU box --
    which is every box this repository has been run on so far: until this test runs somewhere, those legs are unverified on hardware
    (INTEGRATION.md "what has never been run")."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices in one process")
    rng = np.random.default_rng(4242)
    ctxs = [pds.Context(0), pds.Context(1)]
    n, p = 400_000, 5
    X = rng.normal(size=(n, p))
    y = X @ rng.normal(size=p) + 0.3 + 0.1 * rng.normal(size=n)
    cols_dev = [dev(y)] + cols_of(X)
    bounds = [0, n // 2 + 3, n]
    shards = pds.scatter_rows(ctxs, cols_dev, bounds)
    assert all(t.device.index == 1 for t in shards[1])
    for c in range(2):
        assert all(torch.equal(s.cpu(), full[bounds[c]:bounds[c + 1]].cpu()) for s, full in zip(shards[c], cols_dev))
    moms = [pds.gram_moments(*sh[1:], target=sh[0], ctx=ctxs[c], out_device=True).contiguous() for c, sh in enumerate(shards)]
    want = moms[0].cpu() + moms[1].cpu()
    pds.allreduce_sum(ctxs, moms)
    assert all(torch.allclose(m.cpu(), want, rtol=1e-14, atol=0.0) for m in moms)
    b = pds.lin_reg_from_moments(moms[1], add_bias=True, ctx=ctxs[1])
    b = b.cpu().numpy() if hasattr(b, "cpu") else np.asarray(b)
    assert nrel(b.ravel(), orc.pl_lr(X, y, add_bias=True)) < F64_TOL
    blocks = [torch.arange(64 * (c + 1), dtype=torch.float64, device=f"cuda:{c}") + 1000.0 * c for c in range(2)]
    assert torch.equal(pds.gather(ctxs, blocks).cpu(), torch.cat([b_.cpu() for b_ in blocks]))
    # unordered keys over the two devices: per-context moment tables, the peer's table copied across and added, one solve
    G = 3000
    sizes = rng.integers(40, 160, size=G)
    key = np.repeat(np.arange(G, dtype=np.int64) * 2 + 11, sizes)
    N = len(key)
    Xg = rng.normal(size=(N, 4))
    yg = Xg @ rng.normal(size=4) + 1e-3 * key + 0.1 * rng.normal(size=N)
    perm = rng.permutation(N)
    k2, c2, n2 = pds.lin_reg_by_key_multi(*[np.ascontiguousarray(Xg[perm, j]) for j in range(4)], target=yg[perm], key=key[perm], contexts=ctxs)
    k1, c1, n1 = pds.lin_reg_by_key(*[np.ascontiguousarray(Xg[perm, j]) for j in range(4)], target=yg[perm], key=key[perm], ctx=ctxs[0])
    assert np.array_equal(np.asarray(k2), np.asarray(k1)) and np.array_equal(np.asarray(n2), np.asarray(n1))
    assert np.max(np.abs(np.asarray(c2) - np.asarray(c1))) < 1e-9
    for cx in ctxs:
        cx.close()


@pytest.mark.parametrize("n_ctx,n_slices", [(1, 3), (2, 0), (3, 7), (4, 4)])
def test_by_key_multi_context_equals_single_context(pds, orc, n_ctx, n_slices):
    """One host frame through several contexts of ONE process (pds_lr_by_key_multi_*: the route by which a Polars plugin can
    drive several devices; here N contexts on device 0): slices cut at group boundaries, keys and null flags those of the
    single-context call, coefficients equal to rounding -- incl. collinear groups and the rank gate's second pass per slice."""
    rng = np.random.default_rng(500 + 10 * n_ctx + n_slices)
    G, p = 60_000, 5
    sizes = rng.integers(20, 120, size=G)
    keys_g = np.cumsum(rng.integers(1, 4, size=G)) - 7
    key = np.repeat(keys_g, sizes).astype(np.int64)
    N = len(key)
    assert N > 3 * (1 << 20)
    X = rng.normal(size=(N, p))
    y = X @ rng.normal(size=p) + 1e-3 * key + 0.1 * rng.normal(size=N)
    off = np.concatenate([[0], np.cumsum(sizes)])
    for g in range(11, G, 997):
        X[off[g]: off[g + 1], 1] = 2.0 * X[off[g]: off[g + 1], 0]
    cols = [np.ascontiguousarray(X[:, j]) for j in range(p)]
    k1, c1, n1 = pds.lin_reg_by_key(*cols, target=y, key=key, add_bias=True)
    ctxs = [pds.Context(0) for _ in range(n_ctx)]
    k2, c2, n2 = pds.lin_reg_by_key_multi(*cols, target=y, key=key, contexts=ctxs, n_slices=n_slices, add_bias=True)
    assert np.array_equal(k1, k2) and np.array_equal(n1, n2) and n1.sum() >= 50
    # (same kernels, but a slice starts its 128-row tiles at its own first row: a group's rows meet the matrix core in other
    #  tile positions than in the whole-frame call, so the sums round differently -- agreement to rounding, not bit for bit)
    okm = ~n1.astype(bool)
    assert np.max(np.linalg.norm(c1[okm] - c2[okm], axis=1) / np.linalg.norm(c1[okm], axis=1)) < 1e-11
    co_o, nu_o = orc.grouped_lr([y] + cols, off, add_bias=True, nthreads=4)
    assert np.array_equal(n2.astype(bool), nu_o)
    # a capacity that is too small is reported with the total count (the plugin's retry protocol), nothing is written past it
    with pytest.raises(Exception, match="max_groups"):
        pds.lin_reg_by_key_multi(*cols, target=y, key=key, contexts=ctxs, n_slices=n_slices, add_bias=True, max_groups=G - 5)
    # keys out of order (round 4): one row slice per context whatever the order, every context builds the id-indexed moment table
    # of ITS rows (keyed_partition.hip), the tables are summed on the first context, which lists the groups and solves them
    # (SURVEY 8(e) row C3 without moving a row to "its" device).  Against the single-context call AND the oracle at 1e-10.
    perm = rng.permutation(N)
    k3, c3, n3 = pds.lin_reg_by_key_multi(*[c[perm] for c in cols], target=y[perm], key=key[perm], contexts=ctxs, n_slices=n_slices,
                                          add_bias=True)
    from polars_ds_extension_amd import _lib

    assert _lib.load().pds_debug_last_multi_route() == (2 if n_ctx > 1 else 0)  # (the summed-tables route was taken)
    assert np.array_equal(k3, k1) and np.array_equal(n3, n1) and np.array_equal(n3.astype(bool), nu_o)
    ok = ~n1.astype(bool)
    assert np.max(np.linalg.norm(c3[ok] - c1[ok], axis=1) / np.linalg.norm(c1[ok], axis=1)) < 1e-9
    rel_o = np.linalg.norm(c3[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)
    assert np.median(rel_o) < 1e-12 and np.quantile(rel_o, 0.999) < F64_TOL, (np.median(rel_o), rel_o.max())
    with pytest.raises(Exception, match="max_groups"):
        pds.lin_reg_by_key_multi(*[c[perm] for c in cols], target=y[perm], key=key[perm], contexts=ctxs, n_slices=n_slices, add_bias=True,
                                 max_groups=G - 5)
    # per-row predictions over the same contexts (pds_lr_by_key_pred_multi_*: independent slices): the single-context call's rows,
    # with and without weights; a shuffled frame falls back to the single-context route and still lands in frame order
    w = rng.random(N) + 0.25
    for wts in (None, w):
        p1, r1, f1 = pds.lin_reg_by_key_pred(*cols, target=y, key=key, add_bias=True, weights=wts)
        p2, r2, f2 = pds.lin_reg_by_key_pred_multi(*cols, target=y, key=key, contexts=ctxs, n_slices=n_slices, add_bias=True, weights=wts)
        assert np.array_equal(f1, f2) and np.array_equal(np.isnan(p1), np.isnan(p2))
        # (the weighted per-group fit is ungated, as faer_weighted_lr: the collinear groups' rows are NaN without a flag, in both)
        live = ~f1.astype(bool) & np.isfinite(p1)
        scale = np.abs(y[live]) + np.abs(p1[live]) + 1.0
        assert np.max(np.abs(p1[live] - p2[live]) / scale) < 1e-10 and np.max(np.abs(r1[live] - r2[live]) / scale) < 1e-10
        assert np.all(np.isnan(p2[f1.astype(bool)]))
    p3, r3, f3 = pds.lin_reg_by_key_pred_multi(*[c[perm] for c in cols], target=y[perm], key=key[perm], contexts=ctxs, n_slices=n_slices,
                                               add_bias=True)
    p1, r1, f1 = pds.lin_reg_by_key_pred(*cols, target=y, key=key, add_bias=True)
    live = ~f1.astype(bool)
    assert np.array_equal(f3, f1[perm])
    assert np.max(np.abs(p3[live[perm]] - p1[perm][live[perm]]) / (np.abs(y[perm][live[perm]]) + np.abs(p1[perm][live[perm]]) + 1.0)) < 1e-9
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("p,bias,l2", [(20, True, 0.0), (40, True, 0.0), (40, False, 0.2), (100, True, 0.0)])
def test_by_key_shuffled_rows_beyond_16_features(pds, orc, p, bias, l2):
    """Keys in any row order with MORE than 16 features (VERDICT r3 missing 5): the partition route stops at 16, so these
    frames take the stable radix sort + row gather and then the grouped path of their width -- the fused 17 .. 32-feature stream
    (round 4), the record pipeline with the wave-per-system solver (33 .. 64), the tiled SYRK + big-system solver (> 64).  Against the
    oracle on the frame in key order: keys, null flags, coefficients."""
    rng = np.random.default_rng(800 + p)
    pp = p + int(bias)
    G = 700 if p <= 40 else 150
    sizes = rng.integers(3 * pp, 5 * pp, size=G)
    sizes[::53] = rng.integers(1, pp, size=len(sizes[::53]))      # fewer rows than coefficients -> null
    keys_g = (rng.permutation(G) * 5 - 300).astype(np.int64)        # sparse, unordered key values
    key = np.repeat(keys_g, sizes)
    N = len(key)
    X = rng.normal(size=(N, p))
    y = X @ rng.normal(size=p) + 1e-3 * key + 0.1 * rng.normal(size=N) + (0.5 if bias else 0.0)
    off0 = np.concatenate([[0], np.cumsum(sizes)])
    for g in range(9, G, 97):                                        # collinear -> gated (ridge: accepted)
        X[off0[g]: off0[g + 1], 3] = X[off0[g]: off0[g + 1], 0] - X[off0[g]: off0[g + 1], 2]
    perm = rng.permutation(N)
    k1, c1, n1 = pds.lin_reg_by_key(*cols_of(X[perm]), target=dev(y[perm]), key=dev(key[perm]), add_bias=bias, l2_reg=l2)
    k1, c1, n1 = k1.cpu().numpy(), c1.cpu().numpy(), n1.cpu().numpy().astype(bool)
    order = np.argsort(key, kind="stable")
    uk, cnt = np.unique(key[order], return_counts=True)
    assert np.array_equal(k1, uk)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    Xs, ys = X[order], y[order]
    co_o, nu_o = orc.grouped_lr([ys] + [Xs[:, j] for j in range(p)], off, add_bias=bias, l2_reg=l2, nthreads=4)
    assert np.array_equal(n1, nu_o), np.flatnonzero(n1 != nu_o)[:10]
    ok = ~n1
    assert ok.sum() > 0.9 * G and n1.sum() >= G // 53
    err = np.linalg.norm(c1[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)
    for g, e in zip(np.flatnonzero(ok), err):
        if e > F64_TOL:
            Xg = Xs[off[g]: off[g + 1]]
            Xb = np.c_[Xg, np.ones(len(Xg))] if bias else Xg
            assert e < 64 * 2.2e-16 * np.linalg.cond(Xb.T @ Xb + l2 * np.eye(pp)), (g, e)


@pytest.mark.parametrize("p,bias,kw", [(1, False, {}), (3, True, {}), (5, False, {"l2_reg": 0.3}), (8, True, {}), (8, False, {"l1_reg": 0.02}),
                                        (11, False, {}), (16, True, {}), (6, True, {"positive": True})])
def test_by_key_partition_route_against_oracle(pds, orc, p, bias, kw):
    """Shuffled rows, dense integer keys, >= 2^16 rows: the partition route (keyed_partition.hip -- no sort of the rows: bucketed
    records, moments in the accumulate kernel's registers, batched solve) against the oracle per group; sparse key values (every third id unused),
    too-small and collinear groups; and against the sorting route (PDS_KEYED_SORT is read once per process, so the sorted
    call is made on the frame in key order, which moves nothing)."""
    rng = np.random.default_rng(700 + p)
    G = 2500
    sizes = rng.integers(p + 2, 160, size=G)
    sizes[::83] = rng.integers(1, p + 1, size=len(sizes[::83]))  # fewer rows than coefficients -> null
    keys_g = (rng.permutation(G) * 3 - 1000).astype(np.int64)
    key = np.repeat(keys_g, sizes)
    N = len(key)
    assert N >= 1 << 16
    X = rng.normal(size=(N, p))
    y = X @ rng.normal(size=p) + 1e-3 * key + 0.1 * rng.normal(size=N) + (0.5 if bias else 0.0)
    off0 = np.concatenate([[0], np.cumsum(sizes)])
    if p >= 2 and not kw:
        for g in range(7, G, 301):
            X[off0[g]: off0[g + 1], 1] = 2.0 * X[off0[g]: off0[g + 1], 0]
    perm = rng.permutation(N)
    kp, Xp, yp = key[perm], X[perm], y[perm]
    k1, c1, n1 = pds.lin_reg_by_key(*cols_of(Xp), target=dev(yp), key=dev(kp), add_bias=bias, tol=1e-9, max_iter=2000, **kw)
    k1, c1, n1 = k1.cpu().numpy(), c1.cpu().numpy(), n1.cpu().numpy().astype(bool)
    order = np.argsort(key, kind="stable")
    ks, Xs, ys = key[order], X[order], y[order]
    uk, cnt = np.unique(ks, return_counts=True)
    assert np.array_equal(k1, uk)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    pp = p + int(bias)
    worst = 0.0
    nulls_o = np.zeros(len(uk), dtype=bool)
    for g in range(len(uk)):
        sl = slice(off[g], off[g + 1])
        if cnt[g] < pp:
            nulls_o[g] = True
            continue
        b = orc.pl_lr(Xs[sl], ys[sl], add_bias=bias, tol=1e-9, max_iter=2000, **kw)
        if b is None:
            nulls_o[g] = True
            continue
        if cnt[g] >= 2 * pp + 8 and not n1[g]:
            worst = max(worst, np.linalg.norm(c1[g] - b) / max(np.linalg.norm(b), 1e-300))
    assert np.array_equal(n1, nulls_o), (n1.sum(), nulls_o.sum())
    assert worst < (1e-8 if kw.get("l1_reg") or kw.get("positive") else F64_TOL), worst
    # the frame in key order (nothing moves, fused streaming kernel): same groups, same nulls, same fits
    k2, c2, n2 = pds.lin_reg_by_key(*cols_of(Xs), target=dev(ys), key=dev(ks), add_bias=bias, tol=1e-9, max_iter=2000, **kw)
    assert np.array_equal(k2.cpu().numpy(), k1) and np.array_equal(n2.cpu().numpy().astype(bool), n1)
    ok = ~n1 & (cnt >= 2 * pp + 8)
    d = np.linalg.norm(c2.cpu().numpy()[ok] - c1[ok], axis=1) / np.linalg.norm(c1[ok], axis=1)
    assert d.max() < (1e-8 if kw.get("l1_reg") or kw.get("positive") else 1e-9)
    # host frame through the same route
    k3, c3, n3 = pds.lin_reg_by_key(*[np.ascontiguousarray(Xp[:, j]) for j in range(p)], target=yp, key=kp, add_bias=bias, tol=1e-9,
                                    max_iter=2000, **kw)
    assert np.array_equal(k3, k1) and np.array_equal(n3.astype(bool), n1)
    assert np.max(np.linalg.norm(c3[ok] - c1[ok], axis=1) / np.linalg.norm(c1[ok], axis=1)) < 1e-9
    if not kw:
        # per-row predictions of the shuffled frame (grouped_pred.hip MODE 2: the row's group looked up from its key, nothing
        # permuted): the route's own coefficients applied to every row where it lies; rows of null groups are null
        pr, rs, rn = pds.lin_reg_by_key_pred(*cols_of(Xp), target=dev(yp), key=dev(kp), add_bias=bias)
        pr, rs, rn = pr.cpu().numpy(), rs.cpu().numpy(), rn.cpu().numpy().astype(bool)
        gi = np.searchsorted(uk, kp)
        assert np.array_equal(rn, n1[gi])
        Xb = np.c_[Xp, np.ones(N)] if bias else Xp
        live = ~rn
        own = np.einsum("ij,ij->i", Xb[live], c1[gi[live]])
        scale = np.linalg.norm(Xb[live], axis=1) * np.linalg.norm(c1[gi[live]], axis=1)
        assert np.max(np.abs(pr[live] - own) / scale) < 1e-12
        assert np.max(np.abs(rs[live] - (yp[live] - own)) / np.maximum(scale, np.abs(yp[live]))) < 1e-12
        assert np.all(np.isnan(pr[rn])) and np.all(np.isnan(rs[rn]))
        ph, rh, nh = pds.lin_reg_by_key_pred(*[np.ascontiguousarray(Xp[:, j]) for j in range(p)], target=yp, key=kp, add_bias=bias)
        # (not bit for bit: the records of a bucket arrive in the order of the scatter's atomics, so the sums round differently per call)
        assert np.array_equal(nh.astype(bool), rn)
        assert np.max(np.abs(ph[live] - pr[live]) / scale) < 1e-9 and np.max(np.abs(rh[live] - rs[live]) / np.maximum(scale, np.abs(yp[live]))) < 1e-9


def test_by_key_partition_route_unaligned_keys_and_negative_base(pds, orc):
    """The partition route's bucket histogram rides on the order check only for a 16-byte aligned key buffer; an 8-byte aligned one
    takes the separate histogram pass.  Both must give the same groups -- also with negative keys, where the dense ids start at the
    smallest key rounded DOWN to a bucket boundary."""
    import torch

    rng = np.random.default_rng(515)
    G, p = 1800, 3
    sizes = rng.integers(20, 70, size=G)
    keys_g = (rng.permutation(G) * 2 - 2501).astype(np.int64)   # negative and positive, every second id unused
    key = np.repeat(keys_g, sizes)
    N = len(key)
    assert N >= 1 << 16
    X = rng.normal(size=(N, p))
    y = X @ rng.normal(size=p) + 1e-3 * key + 0.1 * rng.normal(size=N)
    perm = rng.permutation(N)
    kp, Xp, yp = key[perm], X[perm], y[perm]
    kbuf = torch.from_numpy(np.concatenate([[0], kp])).cuda()
    k_al = torch.from_numpy(kp).cuda()
    assert kbuf[1:].data_ptr() % 16 == 8 and k_al.data_ptr() % 16 == 0
    k1, c1, n1 = pds.lin_reg_by_key(*cols_of(Xp), target=dev(yp), key=k_al, add_bias=True)
    k2, c2, n2 = pds.lin_reg_by_key(*cols_of(Xp), target=dev(yp), key=kbuf[1:], add_bias=True)
    assert torch.equal(k1, k2) and torch.equal(n1, n2) and np.array_equal(k1.cpu().numpy(), np.sort(keys_g))
    assert float(torch.max(torch.linalg.norm(c1 - c2, dim=1) / torch.linalg.norm(c1, dim=1))) < 1e-10
    order = np.argsort(key, kind="stable")
    off = np.concatenate([[0], np.cumsum(np.unique(key, return_counts=True)[1])]).astype(np.int64)
    co_o, nu_o = orc.grouped_lr([y[order]] + [X[order][:, j] for j in range(p)], off, add_bias=True, nthreads=2)
    assert not nu_o.any() and not n1.any().item()
    assert np.max(np.linalg.norm(c1.cpu().numpy() - co_o, axis=1) / np.linalg.norm(co_o, axis=1)) < 1e-9


def test_by_key_partition_route_f32_and_giant_group(pds, orc, f32):
    """f32 frames (moments in f64 register accumulators) and a skewed frame: one key holds half of the rows, so its bucket is fitted by
    many accumulate workgroups that meet in the table through global atomics."""
    rng = np.random.default_rng(808)
    G, p = 900, 6
    sizes = rng.integers(20, 120, size=G)
    sizes[17] = 120_000
    key = np.repeat(np.arange(G, dtype=np.int64) + 5, sizes)
    N = len(key)
    X = rng.normal(size=(N, p)).astype(np.float32)
    y = (X @ rng.normal(size=p) + 0.1 * rng.normal(size=N) + 0.25).astype(np.float32)
    perm = rng.permutation(N)
    k1, c1, n1 = pds.lin_reg_by_key(*cols_of(X[perm]), target=dev(y[perm]), key=dev(key[perm]), add_bias=True)
    assert c1.dtype.is_floating_point and c1.element_size() == 4 and not n1.any().item()
    c1 = c1.cpu().numpy().astype(np.float64)
    off = np.concatenate([[0], np.cumsum(sizes)])
    for g in (0, 17, 400, G - 1):
        sl = slice(off[g], off[g + 1])
        truth = orc.pl_lr(X[sl].astype(np.float64), y[sl].astype(np.float64), add_bias=True)
        assert np.linalg.norm(c1[g] - truth) / np.linalg.norm(truth) < F32_TOL


def test_grouped_ridge_host_space(pds, orc):
    rng = np.random.default_rng(77)
    G, p = 500, 5
    sizes = rng.integers(20, 200, size=G)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    X = rng.normal(size=(int(off[-1]), p))
    y = X @ rng.normal(size=p) + rng.normal(size=len(X))
    co, nu = pds.lin_reg_by(*[np.ascontiguousarray(X[:, j]) for j in range(p)], target=y, group_offsets=off, add_bias=True,
                            l2_reg=0.2, solver="choleskey")
    co_o, _ = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=True, l2_reg=0.2, solver="choleskey")
    assert isinstance(co, np.ndarray) and not nu.any()
    assert np.max(np.linalg.norm(co - co_o, axis=1) / np.linalg.norm(co_o, axis=1)) < F64_TOL


@pytest.mark.parametrize("p,bias", [(3, True), (8, False), (16, True), (24, False)])
@pytest.mark.parametrize("kw", [dict(l1_reg=0.02), dict(l1_reg=0.01, l2_reg=0.05), dict(positive=True),
                                dict(l2_reg=0.2, positive=True), dict(l1_reg=0.01, positive=True)])
def test_grouped_lasso_elastic_net_nnls(pds, orc, kw, p, bias):
    """group_by(key).agg(pds.lin_reg(l1_reg=..., positive=...)): per group the dispatch of pl_lr (:447-497)."""
    rng = np.random.default_rng(300 + p)
    G = 300
    sizes = rng.integers(p + 10, 200, size=G)
    sizes[::41] = rng.integers(0, p + 1, size=len(sizes[::41]))  # empty / too-small groups -> null
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    X = rng.normal(size=(N, p))
    y = np.empty(N)
    for g in range(G):
        s = slice(off[g], off[g + 1])
        y[s] = X[s] @ rng.normal(size=p) + 0.1 * rng.normal(size=sizes[g]) + (0.7 if bias else 0.0)
    co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias, tol=1e-11, max_iter=5000, **kw)
    co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
    pp = p + bias
    assert np.array_equal(nu, sizes < pp)
    assert np.isnan(co[nu]).all()
    worst = 0.0
    for g in np.flatnonzero(~nu):
        s = slice(off[g], off[g + 1])
        bo = orc.pl_lr(X[s], y[s], add_bias=bias, tol=1e-11, max_iter=5000, **kw)
        worst = max(worst, float(np.linalg.norm(co[g] - bo) / max(np.linalg.norm(bo), 1e-300)))
    assert worst < 1e-8  # both sides iterate to |delta| < 1e-11; the fixed points agree to the conditioning of the group
    # host-space call (numpy in, numpy out) takes the same route
    co_h, nu_h = pds.lin_reg_by(*[np.ascontiguousarray(X[:, j]) for j in range(p)], target=y, group_offsets=off, add_bias=bias,
                                tol=1e-11, max_iter=5000, **kw)
    assert np.array_equal(nu_h.astype(bool), nu) and np.array_equal(co_h[~nu], co[~nu])


@pytest.mark.parametrize("p,bias", [(65, False), (70, True), (130, True)])
def test_grouped_more_than_64_features(pds, orc, p, bias):
    """> 64 features per group: tiled matrix-core Gram build per group + the big-system solver (coverage path)."""
    rng = np.random.default_rng(500 + p)
    pp = p + bias
    sizes = np.array([3 * pp, pp + 40, 0, pp - 1, 2 * pp + 17, 5, 4 * pp, pp + 200])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    X = rng.normal(size=(N, p))
    y = X @ rng.normal(size=p) + (0.7 if bias else 0.0) + 0.1 * rng.normal(size=N)
    for kw in (dict(), dict(l2_reg=0.3), dict(l1_reg=0.01, tol=1e-11, max_iter=5000)):
        co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias, **kw)
        co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
        assert np.isnan(co[nu]).all() and nu[sizes < pp].all() and not nu[sizes >= 3 * pp].any()
        for g in np.flatnonzero(sizes >= pp):
            sl = slice(off[g], off[g + 1])
            bo = orc.pl_lr(X[sl], y[sl], add_bias=bias, **kw)  # None: the rank gate fired (barely more rows than features)
            assert (bo is None) == bool(nu[g])
            if bo is not None:
                assert nrel(co[g], bo) < 1e-8


@pytest.mark.parametrize("p,bias", [(3, True), (16, False)])
def test_grouped_by_key_any_row_order(pds, orc, p, bias):
    """group_by(key) on an unsorted key column: radix sort of (key, row) + gather on the device, then the grouped kernel."""
    rng = np.random.default_rng(700 + p)
    G = 700
    keys_distinct = rng.choice(np.arange(-5000, 5000), size=G, replace=False).astype(np.int64)
    sizes = rng.integers(p + 5, 120, size=G)
    key = np.repeat(keys_distinct, sizes)
    N = len(key)
    X = rng.normal(size=(N, p))
    beta_of = {int(k): rng.normal(size=p) for k in keys_distinct}
    y = np.array([X[i] @ beta_of[int(key[i])] for i in range(N)]) + 0.05 * rng.normal(size=N) + (0.3 if bias else 0.0)
    perm = rng.permutation(N)  # rows in random order
    Xs, ys, ks = X[perm], y[perm], key[perm]
    # reference: the same groups, contiguous (the order of rows inside a group does not matter to a regression)
    order = np.argsort(key, kind="stable")
    ko, off = np.unique(key[order], return_index=True)
    off = np.append(off, N).astype(np.int64)
    co_o, nu_o = orc.grouped_lr([y[order]] + [X[order][:, j] for j in range(p)], off, add_bias=bias, nthreads=4)
    for space in ("device", "host"):
        if space == "device":
            k_out, co, nu = pds.lin_reg_by_key(*cols_of(Xs), target=dev(ys), key=dev(ks), add_bias=bias)
            k_out, co, nu = k_out.cpu().numpy(), co.cpu().numpy(), nu.cpu().numpy()
        else:
            k_out, co, nu = pds.lin_reg_by_key(*[np.ascontiguousarray(Xs[:, j]) for j in range(p)], target=ys, key=ks, add_bias=bias)
        assert np.array_equal(k_out, ko) and np.array_equal(nu.astype(bool), nu_o)
        assert np.max(np.linalg.norm(co - co_o, axis=1) / np.linalg.norm(co_o, axis=1)) < 1e-9
    # keys already in order: no data movement, identical results
    k2, co2, nu2 = pds.lin_reg_by_key(*cols_of(X[order]), target=dev(y[order]), key=dev(key[order]), add_bias=bias)
    assert np.array_equal(k2.cpu().numpy(), ko)
    assert np.max(np.linalg.norm(co2.cpu().numpy() - co_o, axis=1) / np.linalg.norm(co_o, axis=1)) < 1e-9


@pytest.mark.parametrize("shift", [0, 1])
def test_by_key_ordered_keys_run_lengths(pds, shift):
    """Ordered keys: the distinct keys and group offsets come from the order check's run counts + one writing pass (keyed.hip:
    key_run_starts_kernel) -- runs of every length around the 128-key pieces, single-row groups, one long run, tiny frames, and a
    key buffer that is 8- but not 16-byte aligned.  The fits must be those of the offsets entry point on the same groups, bit for bit."""
    import torch

    rng = np.random.default_rng(60 + shift)
    cases = [rng.integers(1, 6, size=700), rng.integers(100, 300, size=40), np.array([1] * 300), np.array([5000]), np.array([3]),
             np.array([1, 1]), np.array([127, 1, 128, 129, 2, 255, 1, 1, 64, 64]), rng.integers(1, 400, size=900),
             # more than 1024 group starts inside one wave's 8192 keys: the writing pass goes round its list several times
             np.array([1] * 20_000), rng.integers(1, 4, size=15_000), np.r_[np.array([1] * 9000), rng.integers(50, 200, size=300)]]
    for sizes in cases:
        keys_g = np.cumsum(rng.integers(1, 9, size=len(sizes))) - 40
        key = np.repeat(keys_g, sizes).astype(np.int64)
        n = len(key)
        x = rng.normal(size=n + shift)
        y = 0.5 * x + 0.1 * rng.normal(size=n + shift)
        kd = torch.from_numpy(np.concatenate([np.zeros(shift, dtype=np.int64), key])).cuda()[shift:]
        xd, yd = torch.from_numpy(x).cuda()[shift:], torch.from_numpy(y).cuda()[shift:]
        k_out, co, nu = pds.lin_reg_by_key(xd, target=yd, key=kd, add_bias=True)
        assert np.array_equal(k_out.cpu().numpy(), keys_g), (len(sizes), n)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        co2, nu2 = pds.lin_reg_by(xd, target=yd, group_offsets=off, add_bias=True)
        assert torch.equal(nu, nu2) and torch.equal(torch.nan_to_num(co), torch.nan_to_num(co2))


def test_by_key_more_groups_than_the_first_guess(pds):
    """Without `max_groups` the outputs are sized for max(2^20, n / 16) groups; an ordered column with more distinct keys than that comes
    back with the count (the order check's run total, before anything is written) and the call repeats with the exact capacity."""
    import torch

    n = 3_000_000
    key = torch.arange(n, dtype=torch.int64, device="cuda") // 2  # 1.5e6 groups of two rows
    x = torch.linspace(0.0, 1.0, n, dtype=torch.float64, device="cuda") + (torch.arange(n, device="cuda") % 2).double()
    y = 2.0 * x + 1.0
    k, co, nu = pds.lin_reg_by_key(x, target=y, key=key, add_bias=True)
    assert k.shape[0] == n // 2 and torch.equal(k, torch.arange(n // 2, dtype=torch.int64, device="cuda"))
    assert int(nu.sum().item()) == 0
    assert float((co[:, 0] - 2.0).abs().max().item()) < 1e-6 and float((co[:, 1] - 1.0).abs().max().item()) < 1e-6


def test_by_key_order_check_sees_every_inversion(pds):
    """keyed.hip's one-pass order check (16-byte loads, the successor of a key from the lane itself, the next lane or the next
    128-key piece): ONE adjacent inversion anywhere -- inside a lane's pair, between lanes, between pieces, in the unaligned head,
    in the tail behind the last whole piece -- must send the frame down the sorting route.  A missed inversion would show: the
    run-length groups of an unsorted key column repeat a key."""
    import torch

    for n, shift in ((128 * 9 + 37, 0), (128 * 9 + 37, 1), (131, 0), (129, 1), (128, 0), (5, 0), (2, 0)):
        base = torch.arange(n + shift, dtype=torch.int64, device="cuda")
        x = torch.linspace(0.0, 1.0, n + shift, dtype=torch.float64, device="cuda")
        spots = sorted({0, 1, 2, 62, 63, 64, 125, 126, 127, 128, 129, 254, 255, 256, 511, 512, n - 3, n - 2} & set(range(n - 1)))
        for i in spots:
            k = base.clone()
            k[shift + i], k[shift + i + 1] = base[shift + i + 1].clone(), base[shift + i].clone()
            kk, xx = k[shift:], x[shift:]  # (shift = 1: a key buffer that is 8- but not 16-byte aligned)
            k_out, _, _ = pds.lin_reg_by_key(xx, target=xx, key=kk)
            assert torch.equal(k_out, base[shift:]), (n, shift, i)
        k_out, _, _ = pds.lin_reg_by_key(x[shift:], target=x[shift:], key=base[shift:])  # ordered: the keys as they are
        assert torch.equal(k_out, base[shift:])


@pytest.mark.parametrize("p,bias", [(24, True), (40, False), (64, True)])
def test_grouped_mid_width_f32(pds, orc, f32, p, bias):
    """f32 frames, 17 .. 64 features per group: f32 Gram records (f64 sums behind f32 products), the wave-per-system solver in f64 on
    them -- the f32 contract (1e-4 normwise) against the f64 truth of the same f32 frame; too-small groups null."""
    rng = np.random.default_rng(777 + p)
    G = 150
    pp = p + int(bias)
    sizes = rng.integers(6 * pp, 10 * pp, size=G)
    sizes[::31] = rng.integers(1, pp, size=len(sizes[::31]))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    X = rng.normal(size=(N, p)).astype(np.float32)
    y = (X.astype(np.float64) @ rng.normal(size=p) + (0.5 if bias else 0.0) + 0.2 * rng.normal(size=N)).astype(np.float32)
    co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias, singular_x_tol=1e-10)
    assert co.dtype.is_floating_point and co.element_size() == 4
    co, nu = co.cpu().numpy().astype(np.float64), nu.cpu().numpy().astype(bool)
    X64, y64 = X.astype(np.float64), y.astype(np.float64)
    co_t, nu_t = orc.grouped_lr([y64] + [X64[:, j] for j in range(p)], off, add_bias=bias, tol=1e-10, nthreads=2)
    assert np.array_equal(nu, nu_t) and nu.sum() >= 4
    ok = ~nu
    assert np.max(np.linalg.norm(co[ok] - co_t[ok], axis=1) / np.linalg.norm(co_t[ok], axis=1)) < F32_TOL


@pytest.mark.parametrize("p", [28, 40, 64])
def test_grouped_stream_record_edges(pds, orc, p):
    """The streamed Gram records of 28 .. 64 features (grouped_mid.hip grouped_mid_stream_kernel): a group that spans many waves (its
    record is summed from partial records by atomics), groups that start / end inside a 4-row step, empty groups, one- and two-row
    groups, a last group that ends in a partial half-tile, and frames shorter than one half-tile -- coefficients and null flags
    against the oracle, group by group."""
    rng = np.random.default_rng(9000 + p)
    for sizes in (np.r_[7000, rng.integers(p + 2, p + 9, size=60), 0, 0, 1, 2, 3 * p + 1, 0, rng.integers(2 * p, 5 * p, size=40), 3, 4 * p + 3],
                  np.r_[2 * p + 1], np.r_[1, 0, 2 * p + 3], np.r_[rng.integers(p + 1, 3 * p, size=7)]):
        sizes = np.asarray(sizes, dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        N, G = int(off[-1]), len(sizes)
        X = rng.normal(size=(N, p))
        y = X @ rng.normal(size=p) + 0.2 + 0.1 * rng.normal(size=N)
        co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=True)
        co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
        co_o, nu_o = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=True, nthreads=2)
        # (groups with barely more rows than columns sit next to the 1e-12 gate: two correct factorisations may disagree there)
        clear = sizes >= 2 * (p + 1)
        assert np.array_equal(nu[clear | (sizes <= p)], nu_o[clear | (sizes <= p)]), (sizes[nu != nu_o], N)
        ok = clear & ~nu
        if ok.any():
            err = np.linalg.norm(co[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)
            assert err.max() < 1e-8, (err.max(), sizes[ok][np.argmax(err)])


@pytest.mark.parametrize("p,bias,l2", [(17, True, 0.0), (24, False, 0.0), (32, True, 0.0), (33, False, 0.3), (40, True, 0.0), (49, True, 0.0),
                                        (57, False, 0.0), (64, True, 0.0), (64, False, 0.2)])
def test_grouped_mid_width_wave_solver(pds, orc, p, bias, l2):
    """17 .. 64 features per group: the Gram records are solved one wave per system in registers (solve_wave.hip: centred L D L',
    rank gate as a pivot-ratio product); collinear groups and groups next to the gate go through the pivoted-QR pass over the marked
    records, groups with fewer rows than coefficients are null.  Against the oracle's gated col-piv QR, group by group."""
    rng = np.random.default_rng(4000 + p)
    G = 260
    pp = p + int(bias)
    # (rows well above the column count: with n ~ p the relative determinant of a random Gram matrix is itself below the 1e-12 gate)
    sizes = rng.integers(5 * pp, 9 * pp, size=G)
    sizes[::37] = rng.integers(1, pp, size=len(sizes[::37]))          # too few rows -> null
    sizes[3::29] = pp + 2                                              # barely enough rows: next to the gate or beyond it
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    X = rng.normal(size=(N, p)) + rng.normal(size=p) * 0.3
    y = X @ rng.normal(size=p) + (0.7 if bias else 0.0) + 0.2 * rng.normal(size=N)
    for g in range(5, G, 41):                                           # collinear groups -> gated
        X[off[g]: off[g + 1], 3] = 2.0 * X[off[g]: off[g + 1], 1] - X[off[g]: off[g + 1], 0]
    for g in range(9, G, 53):                                           # nearly collinear: next to the gate
        X[off[g]: off[g + 1], 2] = X[off[g]: off[g + 1], 1] + 3e-6 * rng.normal(size=int(sizes[g]))
    co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias, l2_reg=l2)
    co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
    co_o, nu_o = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=bias, l2_reg=l2, nthreads=4)
    assert np.array_equal(nu, nu_o), (nu.sum(), nu_o.sum(), np.flatnonzero(nu != nu_o)[:10])
    assert nu.sum() >= 7 and (~nu).sum() >= G // 2
    ok = ~nu
    err = np.linalg.norm(co[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)
    # groups whose own conditioning makes two correct solvers differ by more than 1e-10 (sizes go down to p' + 3 rows): 64 eps cond(X'X)
    worst = 0.0
    for g in np.flatnonzero(ok)[np.argsort(err)[-6:]]:
        Xg = X[off[g]: off[g + 1]]
        Xb = np.c_[Xg, np.ones(len(Xg))] if bias else Xg
        bound = max(F64_TOL, 64 * 2.2e-16 * np.linalg.cond(Xb.T @ Xb + l2 * np.eye(Xb.shape[1])))
        worst = max(worst, float(err[np.flatnonzero(ok) == g][0] / bound))
    assert worst < 1.0, (worst, err.max())
    # solver = "choleskey" IS the register factorisation (no second pass): same accept / reject rule on the clear groups
    co_c, nu_c = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias, l2_reg=l2, solver="choleskey")
    co_c, nu_c = co_c.cpu().numpy(), nu_c.cpu().numpy().astype(bool)
    both = ~nu_c & ok
    assert both.sum() >= 0.8 * ok.sum()
    assert np.median(np.linalg.norm(co_c[both] - co_o[both], axis=1) / np.linalg.norm(co_o[both], axis=1)) < 1e-10


@pytest.mark.parametrize("p,rows,bias", [(64, 98, False), (63, 96, True), (48, 60, False), (32, 34, True)])
def test_grouped_random_groups_at_the_gate(pds, orc, p, rows, bias):
    """Random groups with few more rows than columns: the relative determinant of X'X is ~ 1e-12 by itself (64 columns over 100 rows: the product of
    the pivot ratios n / (n - k) is 6.6e11), so every group sits AT the reference's 1e-12 gate, some on either side -- with no single
    pivot small.  The null decision of the register factorisation must be the reference's (the statistic is the same number: ln det -
    sum ln G_ii does not depend on the pivoting), and only groups within 1e-5 of the threshold may need the pivoted QR."""
    rng = np.random.default_rng(6400 + p)
    G = 3000
    off = np.arange(0, G * rows + 1, rows, dtype=np.int64)
    N = G * rows
    X = rng.normal(size=(N, p))
    y = X @ rng.normal(size=p) + (0.5 if bias else 0.0) + 0.3 * rng.normal(size=N)
    co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias)
    co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
    co_o, nu_o = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=bias, nthreads=4)
    assert np.array_equal(nu, nu_o), (nu.sum(), nu_o.sum(), np.flatnonzero(nu != nu_o)[:10])
    assert 0.02 * G < nu.sum() < 0.98 * G, nu.sum()   # (both sides of the gate are populated: the frame tests what it says)
    ok = ~nu
    err = np.linalg.norm(co[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)
    worst = 0.0
    for g in np.flatnonzero(ok)[np.argsort(err)[-6:]]:
        Xg = X[off[g]: off[g + 1]]
        Xb = np.c_[Xg, np.ones(len(Xg))] if bias else Xg
        bound = max(F64_TOL, 64 * 2.2e-16 * np.linalg.cond(Xb.T @ Xb))
        worst = max(worst, float(err[np.flatnonzero(ok) == g][0] / bound))
    assert worst < 1.0, (worst, err.max())


@pytest.mark.parametrize("p,bias,l2", [(17, True, 0.0), (22, False, 0.05), (28, True, 0.0), (32, False, 0.0)])
def test_grouped_mid_fused_wave_boundaries(pds, orc, p, bias, l2):
    """17 .. 32 f64 features, round 4: ONE stream, a finished group solved in the streaming wave (grouped_mid.hip), no moment
    records.  ~600 waves over this frame: groups cut by wave boundaries (side table + record solver), a group longer than many
    waves' ranges, empty and too-small groups, collinear / nearly collinear groups (marked -> pivoted QR).  Against the oracle."""
    rng = np.random.default_rng(5000 + p)
    pp = p + int(bias)
    sizes = rng.integers(2 * pp, 5 * pp, size=3000)
    sizes[100] = 40_000                      # spans dozens of waves
    sizes[1500] = 3 * 512 + 7                # a few waves
    sizes[::97] = 0                          # empty groups
    sizes[5::113] = rng.integers(1, pp, size=len(sizes[5::113]))  # fewer rows than coefficients -> null
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N, G = int(off[-1]), len(sizes)
    X = rng.normal(size=(N, p)) + rng.normal(size=p) * 0.3
    y = X @ rng.normal(size=p) + (0.7 if bias else 0.0) + 0.2 * rng.normal(size=N)
    for g in range(7, G, 211):               # collinear -> gated
        X[off[g]: off[g + 1], 3] = 2.0 * X[off[g]: off[g + 1], 1] - X[off[g]: off[g + 1], 0]
    for g in range(11, G, 173):              # nearly collinear: next to the gate
        X[off[g]: off[g + 1], 2] = X[off[g]: off[g + 1], 1] + 3e-6 * rng.normal(size=int(sizes[g]))
    co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias, l2_reg=l2)
    co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
    co_o, nu_o = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=bias, l2_reg=l2, nthreads=4)
    assert np.array_equal(nu, nu_o), (nu.sum(), nu_o.sum(), np.flatnonzero(nu != nu_o)[:10])
    ok = ~nu
    assert ok.sum() > 0.9 * G and nu.sum() >= 40
    err = np.linalg.norm(co[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)
    worst = 0.0
    for g in np.flatnonzero(ok)[np.argsort(err)[-6:]]:
        Xg = X[off[g]: off[g + 1]]
        Xb = np.c_[Xg, np.ones(len(Xg))] if bias else Xg
        bound = max(F64_TOL, 64 * 2.2e-16 * np.linalg.cond(Xb.T @ Xb + l2 * np.eye(Xb.shape[1])))
        worst = max(worst, float(err[np.flatnonzero(ok) == g][0] / bound))
    assert worst < 1.0, (worst, err.max())
    assert err[np.flatnonzero(ok) == 100][0] < F64_TOL and err[np.flatnonzero(ok) == 1500][0] < F64_TOL


@pytest.mark.parametrize("p,dtype", [(17, np.float64), (20, np.float64), (26, np.float64), (32, np.float64), (22, np.float32)])
def test_grouped_mid_groups_across_wave_boundaries_repeat_bit_for_bit(pds, p, dtype):
    """Round 6: a group that crosses a wave boundary of the 17 .. 32-feature stream is finished by the wave it starts in (no partial sums
    added by two waves in either order).  ~1000 waves over this frame, nearly every one of them with a group across its end: two calls
    give the same bits, and every group -- the crossing ones included -- equals a plain f64 solve of its rows."""
    rng = np.random.default_rng(6100 + p)
    G = 40_000
    sizes = rng.integers(p + 8, 3 * p, size=G)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    X = (rng.normal(size=(N, p)) + 0.2).astype(dtype)
    y = (X.astype(np.float64) @ rng.normal(size=p) + 0.1 * rng.normal(size=N) + 0.3).astype(dtype)
    if dtype == np.float32:
        pds.config.LIN_REG_EXPR_F64 = False
    try:
        a, na = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=True)
        b, nb = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=True)
    finally:
        pds.config.LIN_REG_EXPR_F64 = True
    import torch
    bits = torch.int64 if dtype == np.float64 else torch.int32  # (null groups carry NaN coefficients: compare the bits)
    assert torch.equal(a.view(bits), b.view(bits)) and torch.equal(na, nb)
    null = na.cpu().numpy().astype(bool)
    assert null.sum() == 0 or dtype == np.float32  # (the f32 default gate is the f32 twin's own)
    a = a.cpu().numpy().astype(np.float64)
    worst = 0.0
    for g in rng.integers(0, G, size=200):
        if null[g]:
            continue
        Xg = np.c_[X[off[g]: off[g + 1]].astype(np.float64), np.ones(int(sizes[g]))]
        beta = np.linalg.lstsq(Xg, y[off[g]: off[g + 1]].astype(np.float64), rcond=None)[0]
        worst = max(worst, float(np.linalg.norm(a[g] - beta) / np.linalg.norm(beta)))
    assert worst < (1e-4 if dtype == np.float32 else 1e-9), worst


@pytest.mark.parametrize("p,bias", [(17, True), (20, False), (24, True), (30, True), (31, True), (32, False)])
def test_grouped_mid_fused_f32_frames(pds, orc, f32, p, bias):
    """f32 frames with 17 .. 32 features take the paired stream too (128-row half-tiles, widened to f64 on their way out of LDS: f64
    moments, f64 solve, f32 coefficients).  Wave boundaries, a group that spans many waves,
    empty and too-small groups; against the f64 truth of the same f32 frame (the f32 contract, 1e-4 normwise)."""
    rng = np.random.default_rng(8100 + p)
    pp = p + int(bias)
    sizes = rng.integers(3 * pp, 6 * pp, size=2500)
    sizes[77] = 50_000
    sizes[::89] = 0
    sizes[3::101] = rng.integers(1, pp, size=len(sizes[3::101]))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N, G = int(off[-1]), len(sizes)
    X = (rng.normal(size=(N, p)) + rng.normal(size=p) * 0.3).astype(np.float32)
    y = (X.astype(np.float64) @ rng.normal(size=p) + (0.5 if bias else 0.0) + 0.2 * rng.normal(size=N)).astype(np.float32)
    co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias, singular_x_tol=1e-10)
    assert co.element_size() == 4
    from polars_ds_extension_amd import _lib
    assert _lib.load().pds_debug_last_grouped_route() == 1
    co, nu = co.cpu().numpy().astype(np.float64), nu.cpu().numpy().astype(bool)
    X64, y64 = X.astype(np.float64), y.astype(np.float64)
    co_t, nu_t = orc.grouped_lr([y64] + [X64[:, j] for j in range(p)], off, add_bias=bias, tol=1e-10, nthreads=4)
    assert np.array_equal(nu, nu_t) and nu.sum() >= 40
    ok = ~nu
    assert np.max(np.linalg.norm(co[ok] - co_t[ok], axis=1) / np.linalg.norm(co_t[ok], axis=1)) < F32_TOL


def test_grouped_mid_fused_falls_back_when_the_marked_list_overflows(pds, orc):
    """More suspect systems than the fused form keeps records for (8192): the record pipeline answers the call.  (Suspect since the rule
    of round 4 = one pivot below 1e-5 of its diagonal: every group carries a nearly collinear pair of columns.)"""
    rng = np.random.default_rng(77)
    p, G = 17, 9000
    sizes = np.full(G, 2 * p + 3)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    X = rng.normal(size=(N, p))
    X[:, 2] = X[:, 1] + 3e-4 * rng.normal(size=N)   # pivot ratio ~ 1e7: suspect, not gated
    y = X @ rng.normal(size=p) + 0.1 * rng.normal(size=N)
    co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=False)
    co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
    co_o, nu_o = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=False, nthreads=4)
    assert np.array_equal(nu, nu_o) and nu.sum() < G // 100
    ok = ~nu
    err = np.linalg.norm(co[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)
    assert np.median(err) < 1e-8   # (cond(X'X) ~ 1e8: two correct solvers differ by that much times eps)
    from polars_ds_extension_amd import _lib
    assert _lib.load().pds_debug_last_grouped_route() == 2, "the marked list did not overflow: the fused form answered"


@pytest.mark.parametrize("p", [5, 20, 40])
def test_grouped_solver_svd_gate(pds, orc, p):
    """group_by().agg(lin_reg(solver="svd")): the streaming kernels answer the clear groups, the groups next to the rank gate go
    through the reference's SVD gate (sum ln sigma_i of X'X against sum ln G_ii, lr_solvers.rs:358-366) -- not through the pivoted
    QR's log-det as before round 4.  Against the oracle's svd route, group by group (tests/test_linear_exprs.py:1326-1340)."""
    rng = np.random.default_rng(600 + p)
    G = 400
    sizes = rng.integers(4 * p, 7 * p, size=G)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    X = rng.normal(size=(N, p))
    y = X @ rng.normal(size=p) + 0.3 + 0.2 * rng.normal(size=N)
    for g in range(3, G, 17):    # collinear -> gated
        X[off[g]: off[g + 1], 2] = X[off[g]: off[g + 1], 0] - 2.0 * X[off[g]: off[g + 1], 1]
    for k, g in enumerate(range(5, G, 13)):    # nearly collinear, on both sides of the 1e-12 gate
        X[off[g]: off[g + 1], 3] = X[off[g]: off[g + 1], 1] + 10.0 ** (-4 - (k % 5)) * rng.normal(size=int(sizes[g]))
    co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=True, solver="svd")
    co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
    co_o, nu_o = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=True, solver="svd", nthreads=4)
    assert np.array_equal(nu, nu_o), np.flatnonzero(nu != nu_o)[:10]
    assert nu.sum() >= G // 17 and (~nu).sum() > G // 2
    ok = ~nu
    err = np.linalg.norm(co[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)
    for g, e in zip(np.flatnonzero(ok), err):
        if e > F64_TOL:
            Xb = np.c_[X[off[g]: off[g + 1]], np.ones(int(sizes[g]))]
            assert e < 64 * 2.2e-16 * np.linalg.cond(Xb.T @ Xb), (g, e)
    # unordered dense keys with "svd": the sorting route (the partition route has no marked pass), same answers
    if p == 5:
        key = np.repeat(np.arange(G, dtype=np.int64), sizes)
        perm = rng.permutation(N)
        k2, co2, nu2 = pds.lin_reg_by_key(*cols_of(X[perm]), target=dev(y[perm]), key=dev(key[perm]), add_bias=True, solver="svd")
        assert np.array_equal(nu2.cpu().numpy().astype(bool), nu_o)
        co2 = co2.cpu().numpy()
        assert np.median(np.linalg.norm(co2[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)) < 1e-10


@pytest.mark.parametrize("p,bias", [(4, True), (15, True), (16, False), (20, True)])
def test_grouped_weighted(pds, orc, p, bias):
    """group_by(key).agg(pds.lin_reg(..., weights=w)): per group faer_weighted_lr (lr_solvers.rs:386-409)."""
    rng = np.random.default_rng(900 + p)
    G = 200
    sizes = rng.integers(p + 8, 150, size=G)
    sizes[::37] = rng.integers(0, p + 1, size=len(sizes[::37]))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    X = rng.normal(size=(N, p))
    y = X @ rng.normal(size=p) + (0.4 if bias else 0.0) + 0.2 * rng.normal(size=N)
    w = rng.random(N) + 0.1
    pp = p + bias
    for space in ("device", "host"):
        if space == "device":
            co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias, weights=dev(w))
            co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
        else:
            co, nu = pds.lin_reg_by(*[np.ascontiguousarray(X[:, j]) for j in range(p)], target=y, group_offsets=off, add_bias=bias,
                                    weights=w)
            nu = nu.astype(bool)
        assert np.array_equal(nu, sizes < pp)
        for g in np.flatnonzero(~nu)[::9]:
            s = slice(off[g], off[g + 1])
            assert nrel(co[g], orc.pl_lr(X[s], y[s], add_bias=bias, weights=w[s])) < 1e-9


def test_grouped_by_key_more_groups_than_first_guess(pds):
    """lin_reg_by_key sizes its outputs for n_rows / 16 keys and repeats with the exact count when there are more."""
    n = (1 << 21) + 12_345
    rng = np.random.default_rng(1)
    key = rng.permutation(n).astype(np.int64) // 2  # ~1.05e6 + distinct keys, two rows each, shuffled
    x = rng.normal(size=n)
    y = 3.0 * x
    k, co, nu = pds.lin_reg_by_key(dev(x), target=dev(y), key=dev(key))
    k, co, nu = k.cpu().numpy(), co.cpu().numpy(), nu.cpu().numpy().astype(bool)
    assert len(k) == (n + 1) // 2 and np.array_equal(k, np.arange(len(k)))
    ok = ~nu
    assert ok.sum() > len(k) - 2 and np.allclose(co[ok, 0], 3.0, rtol=1e-9)


# ------------------------------------------------------------------------------------------ rolling / recursive
def test_rolling_golden_notebook(pds, golden):
    for part in ("rolling_w5_head", "rolling_w5_tail"):
        rows = golden[part]
        X = np.array([[r["x1"], r["x2"]] for r in rows])
        y = np.array([r["y"] for r in rows])
        co, pr, va = pds.rolling_lin_reg(*cols_of(X), target=dev(y), window_size=5)
        co, pr, va = co.cpu().numpy(), pr.cpu().numpy(), va.cpu().numpy()
        assert list(va) == [0, 0, 0, 0, 1]
        np.testing.assert_allclose(co[4], rows[4]["coeffs"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(pr[4], rows[4]["pred"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("w", [5, 8, 12, 15, 64, 256, 300])
@pytest.mark.parametrize("bias,lam", [(False, 0.0), (True, 0.0), (False, 0.1)])
def test_rolling_vs_oracle(pds, orc, w, bias, lam):
    rng = np.random.default_rng(w)
    n, p = 20_000, 3
    X = rng.random((n, p))
    y = X @ [0.2, 0.3, -0.1] + (0.4 if bias else 0.0) + rng.normal(size=n) * 0.1
    co, pr, va = pds.rolling_lin_reg(*cols_of(X), target=dev(y), window_size=w, add_bias=bias, l2_reg=lam)
    co, pr, va = co.cpu().numpy(), pr.cpu().numpy(), va.cpu().numpy()
    Xb = np.c_[X, np.ones(n)] if bias else X
    ref = orc.rolling_lr(Xb, y, w, lam)  # the reference's sequential Woodbury recursion
    assert va[: w - 1].sum() == 0 and va[w - 1 :].all()
    # the reference's own pin is "rolling == per-window lin_reg" at rel 1e-5 / abs 1e-8 (test_rolling_lin_reg);
    # the Woodbury chain itself drifts ~1e-13..1e-10 from the direct solve, so the bar here is 1e-8 normwise
    # against the chain and 1e-10 against the direct per-window solve.
    err_chain = np.linalg.norm(co[w - 1 :] - ref, axis=1) / np.linalg.norm(ref, axis=1)
    # tiny windows (5 rows, 4 coefficients) are routinely near-singular: every such window permanently damages the
    # never-re-anchored chain (its error grows from 1e-15 to 1e-5 over 20k rows here), so for them the chain is
    # only compared before the damage accumulates; the direct per-window solve below is the arbiter throughout.
    assert np.max(err_chain[:50]) < 1e-8 and (w < 64 or np.max(err_chain) < 1e-8)
    idx = rng.integers(w - 1, n, size=200)
    for i in idx:
        A = Xb[i - w + 1 : i + 1]
        direct = np.linalg.solve(A.T @ A + lam * np.eye(A.shape[1]), A.T @ y[i - w + 1 : i + 1])
        cond = np.linalg.cond(A.T @ A + lam * np.eye(A.shape[1]))
        assert nrel(co[i], direct) < max(F64_TOL, 1e-15 * cond)
        assert abs(pr[i] - Xb[i] @ direct) < 1e-9 * max(1.0, cond * 1e-6)


def test_rolling_long_no_drift(pds):
    # 5e6 rows, window 256: tile anchoring keeps the error flat (the reference's chain never re-anchors)
    rng = np.random.default_rng(3)
    n, p, w = 5_000_000, 8, 256
    X = rng.random((n, p))
    y = X @ rng.normal(size=p) + 1e-3 * rng.normal(size=n)
    co, pr, va = pds.rolling_lin_reg(*cols_of(X), target=dev(y), window_size=w)
    co = co.cpu().numpy()
    for i in (w - 1, 4095, 4096, 4097, 1_000_000, n - 1):
        A = X[i - w + 1 : i + 1]
        direct = np.linalg.lstsq(A, y[i - w + 1 : i + 1], rcond=None)[0]
        assert nrel(co[i], direct) < F64_TOL


def test_recursive_vs_oracle(pds, orc):
    rng = np.random.default_rng(11)
    n, p = 30_000, 4
    X = rng.random((n, p))
    y = X @ [0.2, 0.3, -0.1, 0.7] + rng.normal(size=n) * 0.1
    for bias, lam, n0 in ((False, 0.0, 6), (True, 0.0, 10), (False, 0.05, 4)):
        co, pr, va = pds.recursive_lin_reg(*cols_of(X), target=dev(y), start_with=n0, add_bias=bias, l2_reg=lam)
        co, va = co.cpu().numpy(), va.cpu().numpy()
        Xb = np.c_[X, np.ones(n)] if bias else X
        ref = orc.recursive_lr(Xb, y, n0, lam)
        assert va[: n0 - 1].sum() == 0 and va[n0 - 1 :].all()
        late = slice(n0 - 1 + 50, None)  # the first few fits are near-singular: compare where cond is sane
        err = np.linalg.norm(co[late] - ref[50:], axis=1) / np.linalg.norm(ref[50:], axis=1)
        assert np.max(err) < 1e-8
        for i in (200, 5000, n - 1):
            direct = np.linalg.solve(Xb[: i + 1].T @ Xb[: i + 1] + lam * np.eye(Xb.shape[1]), Xb[: i + 1].T @ y[: i + 1])
            assert nrel(co[i], direct) < F64_TOL


@pytest.mark.parametrize("bias,lam", [(True, 0.0), (False, 0.0), (True, 0.03)])
def test_recursive_seeded_continues_a_frame(pds, bias, lam):
    # row-sharded expanding fit = local fit seeded with the moment matrix of the rows in front of the shard
    rng = np.random.default_rng(21)
    n, p = 40_000, 5
    X = rng.random((n, p))
    y = X @ rng.normal(size=p) + 0.4 + rng.normal(size=n) * 0.1
    n0 = 9
    co, pr, va = pds.recursive_lin_reg(*cols_of(X), target=dev(y), start_with=n0, add_bias=bias, l2_reg=lam)
    co, pr, va = co.cpu().numpy(), pr.cpu().numpy(), va.cpu().numpy()
    for k in (4, 13_337):  # cut before start_with is reached / deep inside
        seed = pds.gram_moments(*cols_of(X[:k]), target=dev(y[:k]))
        co2, pr2, va2 = pds.recursive_lin_reg(*cols_of(X[k:]), target=dev(y[k:]), start_with=n0, add_bias=bias, l2_reg=lam,
                                              seed_moments=seed)
        co2, pr2, va2 = co2.cpu().numpy(), pr2.cpu().numpy(), va2.cpu().numpy()
        assert np.array_equal(va2, va[k:])
        ok = va[k:].astype(bool)
        ok[: max(0, 60 - k)] = False  # the very first fits are near-singular
        err = np.linalg.norm(co2[ok] - co[k:][ok], axis=1) / np.linalg.norm(co[k:][ok], axis=1)
        assert np.max(err) < 1e-9
        assert np.max(np.abs(pr2[ok] - pr[k:][ok])) < 1e-9
    # a seed that is not the (p+2)^2 block is refused
    with pytest.raises(ValueError):
        pds.recursive_lin_reg(*cols_of(X), target=dev(y), start_with=n0, seed_moments=np.zeros((3, 3)))


def test_rolling_halo_shards_equal_single_frame(pds):
    # rolling shards by row range with a (window - 1)-row halo; no collective
    rng = np.random.default_rng(22)
    n, p, w = 50_000, 4, 128
    X = rng.random((n, p))
    y = X @ rng.normal(size=p) - 0.2 + rng.normal(size=n) * 0.05
    co, pr, va = pds.rolling_lin_reg(*cols_of(X), target=dev(y), window_size=w, add_bias=True)
    co, pr, va = co.cpu().numpy(), pr.cpu().numpy(), va.cpu().numpy()
    from polars_ds_extension_amd import parallel as par

    for world in (2, 3):
        for r in range(world):
            lo, hi = par.shard_bounds(n, world, r)
            h0 = max(0, lo - (w - 1))
            c2, p2, v2 = pds.rolling_lin_reg(*cols_of(X[h0:hi]), target=dev(y[h0:hi]), window_size=w, add_bias=True)
            c2, p2, v2 = c2.cpu().numpy()[lo - h0 :], p2.cpu().numpy()[lo - h0 :], v2.cpu().numpy()[lo - h0 :]
            assert np.array_equal(v2, va[lo:hi])
            ok = v2.astype(bool)
            assert np.max(np.abs(c2[ok] - co[lo:hi][ok])) < 1e-9 and np.max(np.abs(p2[ok] - pr[lo:hi][ok])) < 1e-9


@pytest.mark.parametrize("p,bias", [(8, True), (9, False), (10, False), (10, True), (11, True), (12, False)])
def test_rolling_recursive_many_coefficients(pds, orc, p, bias):
    # p' >= 10 carries more than 64 running moments per wave (two per lane)
    rng = np.random.default_rng(40 + p)
    n, w, n0 = 6000, 64, 40
    X = rng.random((n, p))
    y = X @ rng.normal(size=p) + 0.3 + 0.05 * rng.normal(size=n)
    Xb = np.c_[X, np.ones(n)] if bias else X
    co, pr, va = pds.rolling_lin_reg(*cols_of(X), target=dev(y), window_size=w, add_bias=bias)
    ref = orc.rolling_lr(Xb, y, w)
    co = co.cpu().numpy()
    assert np.max(np.linalg.norm(co[w - 1 :] - ref, axis=1) / np.linalg.norm(ref, axis=1)) < 1e-9
    assert np.max(np.abs(pr.cpu().numpy()[w - 1 :] - np.einsum("ij,ij->i", Xb[w - 1 :], ref))) < 1e-9
    co2, _, va2 = pds.recursive_lin_reg(*cols_of(X), target=dev(y), start_with=n0, add_bias=bias)
    ref2 = orc.recursive_lr(Xb, y, n0)
    assert va2.cpu().numpy()[n0 - 1 :].all()
    late = 100  # the first fits are near-singular
    err = np.linalg.norm(co2.cpu().numpy()[n0 - 1 + late :] - ref2[late:], axis=1) / np.linalg.norm(ref2[late:], axis=1)
    assert np.max(err) < 1e-9


@pytest.mark.parametrize("p,bias,w", [(13, False, 64), (15, True, 100), (16, True, 257), (20, True, 300), (33, False, 128),
                                      (63, True, 200), (64, False, 500)])
def test_rolling_recursive_13_to_64_coefficients(pds, orc, p, bias, w):
    """p' > 12: per-row moment records + the batched pivoted QR (rolling_wide.hip); crosses segment (256 rows) borders."""
    rng = np.random.default_rng(140 + p)
    n, n0 = 2500, 2 * p + 30
    X = rng.random((n, p))
    y = X @ rng.normal(size=p) + 0.3 + 0.05 * rng.normal(size=n)
    Xb = np.c_[X, np.ones(n)] if bias else X
    for lam in (0.0, 0.05):
        co, pr, va = pds.rolling_lin_reg(*cols_of(X), target=dev(y), window_size=w, add_bias=bias, l2_reg=lam)
        co, va = co.cpu().numpy(), va.cpu().numpy().astype(bool)
        assert va[w - 1 :].all() and not va[: w - 1].any() and np.isnan(co[: w - 1]).all()
        for i in list(range(w - 1, n, 97)) + [n - 1]:  # direct window fits (lambda on every diagonal)
            Z, t = Xb[i - w + 1 : i + 1], y[i - w + 1 : i + 1]
            direct = np.linalg.solve(Z.T @ Z + lam * np.eye(Z.shape[1]), Z.T @ t)
            assert nrel(co[i], direct) < 1e-8
            assert abs(float(pr[i]) - Xb[i] @ direct) < 1e-8
    ref = orc.rolling_lr(Xb, y, w)  # the reference's Woodbury chain
    co, _, _ = pds.rolling_lin_reg(*cols_of(X), target=dev(y), window_size=w, add_bias=bias)
    assert np.max(np.linalg.norm(co.cpu().numpy()[w - 1 :] - ref, axis=1) / np.linalg.norm(ref, axis=1)) < 1e-7
    co2, pr2, va2 = pds.recursive_lin_reg(*cols_of(X), target=dev(y), start_with=n0, add_bias=bias)
    co2, va2 = co2.cpu().numpy(), va2.cpu().numpy().astype(bool)
    assert va2[n0 - 1 :].all() and not va2[: n0 - 1].any()
    for i in (n0 + 50, 255, 256, 257, 1000, n - 1):
        direct = np.linalg.lstsq(Xb[: i + 1], y[: i + 1], rcond=None)[0]
        assert nrel(co2[i], direct) < 1e-8
    # seeded continuation (row-sharded expanding fit): second half seeded with the first half's moments
    h = 1111
    M = pds.gram_moments(*cols_of(X[:h]), target=dev(y[:h]))
    co3, _, va3 = pds.recursive_lin_reg(*cols_of(X[h:]), target=dev(y[h:]), start_with=n0, add_bias=bias, seed_moments=M)
    assert va3.cpu().numpy().all()
    assert np.max(np.linalg.norm(co3.cpu().numpy() - co2[h:], axis=1) / np.linalg.norm(co2[h:], axis=1)) < 1e-9
    # skipping variant: non-finite rows are left out, min_valid_rows on the finite-row count
    Xn = X.copy()
    bad = rng.choice(n, size=60, replace=False)
    Xn[bad, 1] = np.nan
    m = w - 3
    co4, _, va4 = pds.rolling_lin_reg(*cols_of(Xn), target=dev(y), window_size=w, add_bias=bias, skip_non_finite=True,
                                      min_valid_rows=m)
    co4, va4 = co4.cpu().numpy(), va4.cpu().numpy().astype(bool)
    fin = np.isfinite(Xn).all(axis=1)
    cnt = np.convolve(fin.astype(int), np.ones(w, dtype=int))[w - 1 : n]
    assert np.array_equal(va4[w - 1 :], cnt >= m) and not va4[: w - 1].any()
    for i in np.flatnonzero(va4)[::211]:
        rows = np.arange(i - w + 1, i + 1)[fin[i - w + 1 : i + 1]]
        direct = np.linalg.lstsq(Xb[rows], y[rows], rcond=None)[0]
        if fin[i]:
            assert nrel(co4[i], direct) < 1e-8


@pytest.mark.parametrize("p,bias,w", [(65, False, 150), (70, True, 200), (127, True, 300), (130, False, 400)])
def test_rolling_recursive_beyond_64_coefficients(pds, orc, p, bias, w):
    """p' > 64 (the reference's drivers have no limit, lr_online_solvers.rs:148-301): 1024-thread record kernel + the
    big-system solver; crosses segment (256 rows) and record-chunk borders; rolling, skipping, expanding, seeded."""
    rng = np.random.default_rng(900 + p)
    n, n0 = 1500, 2 * p + 30
    X = rng.random((n, p))
    y = X @ rng.normal(size=p) + 0.3 + 0.05 * rng.normal(size=n)
    Xb = np.c_[X, np.ones(n)] if bias else X
    for lam in (0.0, 0.05):
        co, pr, va = pds.rolling_lin_reg(*cols_of(X), target=dev(y), window_size=w, add_bias=bias, l2_reg=lam)
        co, va = co.cpu().numpy(), va.cpu().numpy().astype(bool)
        assert va[w - 1 :].all() and not va[: w - 1].any() and np.isnan(co[: w - 1]).all()
        for i in list(range(w - 1, n, 173)) + [255 + w, 256 + w, n - 1]:
            Z, t = Xb[i - w + 1 : i + 1], y[i - w + 1 : i + 1]
            direct = np.linalg.solve(Z.T @ Z + lam * np.eye(Z.shape[1]), Z.T @ t)
            assert nrel(co[i], direct) < 1e-7, (i, lam)
            assert abs(float(pr[i]) - Xb[i] @ direct) < 1e-7
    co2, pr2, va2 = pds.recursive_lin_reg(*cols_of(X), target=dev(y), start_with=n0, add_bias=bias)
    co2, va2 = co2.cpu().numpy(), va2.cpu().numpy().astype(bool)
    assert va2[n0 - 1 :].all() and not va2[: n0 - 1].any()
    for i in (n0 + 50, 511, 512, 513, 1000, n - 1):
        direct = np.linalg.lstsq(Xb[: i + 1], y[: i + 1], rcond=None)[0]
        assert nrel(co2[i], direct) < 1e-7
    ref = orc.recursive_lr(Xb, y, n0)  # the reference's Woodbury chain from its initial fit
    assert np.max(np.linalg.norm(co2[n0 - 1 :] - ref, axis=1) / np.linalg.norm(ref, axis=1)) < 1e-6
    h = 777
    M = pds.gram_moments(*cols_of(X[:h]), target=dev(y[:h]))
    co3, _, va3 = pds.recursive_lin_reg(*cols_of(X[h:]), target=dev(y[h:]), start_with=n0, add_bias=bias, seed_moments=M)
    assert va3.cpu().numpy().all()
    assert np.max(np.linalg.norm(co3.cpu().numpy() - co2[h:], axis=1) / np.linalg.norm(co2[h:], axis=1)) < 1e-8
    Xn = X.copy()
    bad = rng.choice(n, size=40, replace=False)
    Xn[bad, 1] = np.nan
    m = w - 3
    co4, _, va4 = pds.rolling_lin_reg(*cols_of(Xn), target=dev(y), window_size=w, add_bias=bias, skip_non_finite=True,
                                      min_valid_rows=m)
    co4, va4 = co4.cpu().numpy(), va4.cpu().numpy().astype(bool)
    fin = np.isfinite(Xn).all(axis=1)
    cnt = np.convolve(fin.astype(int), np.ones(w, dtype=int))[w - 1 : n]
    assert np.array_equal(va4[w - 1 :], cnt >= m) and not va4[: w - 1].any()
    for i in np.flatnonzero(va4)[::301]:
        rows = np.arange(i - w + 1, i + 1)[fin[i - w + 1 : i + 1]]
        direct = np.linalg.lstsq(Xb[rows], y[rows], rcond=None)[0]
        if fin[i]:
            assert nrel(co4[i], direct) < 1e-7


def test_rolling_wide_many_chunks(pds):
    """rolling_wide.hip streams its per-row moment records in chunks sized for the Infinity Cache: cross the chunk borders."""
    rng = np.random.default_rng(5)
    n, p, w = 13_000, 63, 150  # 65^2 doubles per row -> ~5.9k rows per chunk
    X = rng.random((n, p))
    y = X @ rng.normal(size=p) + 0.2 + 0.05 * rng.normal(size=n)
    Xb = np.c_[X, np.ones(n)]
    co, pr, va = pds.rolling_lin_reg(*cols_of(X), target=dev(y), window_size=w, add_bias=True)
    co, va = co.cpu().numpy(), va.cpu().numpy().astype(bool)
    assert va[w - 1 :].all() and not va[: w - 1].any()
    for i in (w - 1, 5887, 5888, 5889, 6100, 11775, 11776, 11777, n - 1):
        direct = np.linalg.lstsq(Xb[i - w + 1 : i + 1], y[i - w + 1 : i + 1], rcond=None)[0]
        assert nrel(co[i], direct) < 1e-7
        assert abs(float(pr[i]) - Xb[i] @ direct) < 1e-7
    co2, _, va2 = pds.recursive_lin_reg(*cols_of(X), target=dev(y), start_with=200, add_bias=True)
    co2 = co2.cpu().numpy()
    for i in (199, 5888, 11776, n - 1):
        direct = np.linalg.lstsq(Xb[: i + 1], y[: i + 1], rcond=None)[0]
        assert nrel(co2[i], direct) < 1e-7


def test_rolling_skip_non_finite(pds, orc):
    rng = np.random.default_rng(9)
    n = 3000
    X = rng.random((n, 2))
    y = X @ [1.0, -1.0] + rng.random(n) * 0.01
    bad = rng.choice(n, size=150, replace=False)
    X[bad, 0] = np.nan
    w, m = 10, 6
    co, pr, va = pds.rolling_lin_reg(*cols_of(X), target=dev(y), window_size=w, skip_non_finite=True, min_valid_rows=m)
    co, va = co.cpu().numpy(), va.cpu().numpy().astype(bool)
    ref, valid = orc.rolling_skipping_lr(X, y, w, m)
    assert np.array_equal(va[w - 1 :], valid) and not va[: w - 1].any()  # tests/test_linear_exprs.py:858-908
    fin = np.isfinite(X).all(axis=1)
    for i in np.flatnonzero(va)[::7]:
        rows = np.arange(i - w + 1, i + 1)[fin[i - w + 1 : i + 1]]
        direct = np.linalg.lstsq(X[rows], y[rows], rcond=None)[0]
        assert nrel(co[i], direct) < 1e-9


@pytest.mark.parametrize("pp,bias", [(2, False), (2, True), (4, True), (6, False), (8, False), (8, True)])
@pytest.mark.parametrize("n,w", [(33, 5), (4099, 31), (4096 * 4 + 5, 32), (40_013, 256), (70_001, 1000)])
def test_rolling_even_widths_frame_granules(pds, pp, bias, n, w):
    """Rolling + expanding fits at even p' with ridge: frame lengths off every granule of the kernels (2-row pieces, 4-row chains, 32- and
    256-row stages, 4096- / 16384-row tiles), windows shorter / longer than a stage and a tile, direct window solves at the granule edges.
    (Written for round 4's two-lanes-per-chain kernel -- tools/experiments/rolling_pair_dev.hpp.txt, measured slower and parked --; the
    shapes now run against the product kernel.)"""
    if w < pp:
        pytest.skip("window shorter than the coefficient count")
    rng = np.random.default_rng(1000 * pp + w)
    p = pp - (1 if bias else 0)
    X = rng.random((n, p))
    y = X @ rng.normal(size=p) + (0.3 if bias else 0.0) + 0.05 * rng.normal(size=n)
    Xb = np.c_[X, np.ones(n)] if bias else X
    lam = 0.01
    co, pr, va = pds.rolling_lin_reg(*cols_of(X), target=dev(y), window_size=w, add_bias=bias, l2_reg=lam)
    co, pr, va = co.cpu().numpy(), pr.cpu().numpy(), va.cpu().numpy().astype(bool)
    assert va[w - 1 :].all() and not va[: w - 1].any()
    assert np.isnan(co[: w - 1]).all() and np.isnan(pr[: w - 1]).all() and np.isfinite(co[w - 1 :]).all()
    rows = np.unique(np.r_[w - 1, w, n - 1, n - 2, rng.integers(w - 1, n, size=150),
                           [r for r in (31, 32, 33, 127, 128, 4095, 4096, 4097, 8191, 8192, 16383, 16384, 16385) if w - 1 <= r < n]]).astype(int)
    for i in rows:
        A = Xb[i - w + 1 : i + 1]
        direct = np.linalg.solve(A.T @ A + lam * np.eye(pp), A.T @ y[i - w + 1 : i + 1])
        assert nrel(co[i], direct) < F64_TOL, i
        assert abs(pr[i] - Xb[i] @ direct) < 1e-9
    # expanding fit over the same frame
    n0 = max(pp + 2, 7)
    co2, pr2, va2 = pds.recursive_lin_reg(*cols_of(X), target=dev(y), start_with=n0, add_bias=bias, l2_reg=lam)
    co2, pr2, va2 = co2.cpu().numpy(), pr2.cpu().numpy(), va2.cpu().numpy().astype(bool)
    assert va2[n0 - 1 :].all() and not va2[: n0 - 1].any()
    for i in np.unique(np.r_[n0 - 1, n - 1, rng.integers(n0 - 1, n, size=40), [r for r in (4095, 4096, 16384) if n0 - 1 <= r < n]]).astype(int):
        direct = np.linalg.solve(Xb[: i + 1].T @ Xb[: i + 1] + lam * np.eye(pp), Xb[: i + 1].T @ y[: i + 1])
        assert nrel(co2[i], direct) < 1e-9, i
        assert abs(pr2[i] - Xb[i] @ direct) < 1e-8


@pytest.mark.parametrize("pp,bias,w,m", [(2, False, 10, 6), (4, True, 40, 20), (8, False, 256, 200)])
def test_rolling_even_widths_non_finite_rows(pds, orc, pp, bias, w, m):
    """Non-finite rows at even p': left out of the sums, counted out of the window, NaN pred (lr_online_solvers.rs:85-89, 218-301)."""
    rng = np.random.default_rng(17 + pp)
    n = 9000
    p = pp - (1 if bias else 0)
    X = rng.random((n, p))
    y = X @ rng.normal(size=p) + rng.random(n) * 0.01
    bad = rng.choice(n, size=300, replace=False)
    X[bad[:200], rng.integers(0, p, size=200)] = np.nan
    y[bad[200:250]] = np.inf
    X[bad[250:], 0] = -np.inf
    Xb = np.c_[X, np.ones(n)] if bias else X
    co, pr, va = pds.rolling_lin_reg(*cols_of(X), target=dev(y), window_size=w, add_bias=bias, skip_non_finite=True, min_valid_rows=m)
    co, pr, va = co.cpu().numpy(), pr.cpu().numpy(), va.cpu().numpy().astype(bool)
    ref, valid = orc.rolling_skipping_lr(Xb, y, w, m)
    assert np.array_equal(va[w - 1 :], valid) and not va[: w - 1].any()
    fin = np.isfinite(Xb).all(axis=1) & np.isfinite(y)
    for i in np.flatnonzero(va)[::23]:
        rows = np.arange(i - w + 1, i + 1)[fin[i - w + 1 : i + 1]]
        direct = np.linalg.lstsq(Xb[rows], y[rows], rcond=None)[0]
        assert nrel(co[i], direct) < 1e-8, i
        assert np.isnan(pr[i]) if not fin[i] else abs(pr[i] - Xb[i] @ direct) < 1e-8


def test_new_paths_f32(pds, orc, f32):
    """f32 twins of the coverage paths: wide rolling, keyed grouping, grouped lasso, grouped > 64 features, HC3 beyond 16."""
    rng = np.random.default_rng(77)
    n, p, w = 3000, 20, 200
    X = rng.random((n, p)).astype(np.float32)
    y = (X @ rng.normal(size=p) + 0.3 + 0.05 * rng.normal(size=n)).astype(np.float32)
    co, pr, va = pds.rolling_lin_reg(*cols_of(X), target=dev(y), window_size=w, add_bias=True)
    co = co.cpu().numpy()
    assert co.dtype == np.float32 and va.cpu().numpy()[w - 1 :].all()
    Xb = np.c_[X.astype(np.float64), np.ones(n)]
    for i in (w - 1, 1500, n - 1):
        direct = np.linalg.lstsq(Xb[i - w + 1 : i + 1], y[i - w + 1 : i + 1].astype(np.float64), rcond=None)[0]
        assert nrel(co[i], direct) < F32_TOL  # (measured 4e-6: tools/f32_distances.py)
    # keyed grouping, shuffled rows
    G, per, q = 300, 60, 4
    key = np.repeat(np.arange(G) * 7 - 100, per)
    Xg = rng.normal(size=(G * per, q)).astype(np.float32)
    yg = (Xg @ rng.normal(size=q) + 0.1 * rng.normal(size=G * per)).astype(np.float32)
    perm = rng.permutation(G * per)
    k, cg, nu = pds.lin_reg_by_key(*cols_of(Xg[perm]), target=dev(yg[perm]), key=dev(key[perm]))
    assert np.array_equal(k.cpu().numpy(), np.arange(G) * 7 - 100) and not nu.cpu().numpy().any()
    cg = cg.cpu().numpy()
    for g in (0, 123, G - 1):
        m = key == g * 7 - 100
        assert nrel(cg[g], np.linalg.lstsq(Xg[m].astype(np.float64), yg[m].astype(np.float64), rcond=None)[0]) < F32_TOL
    # grouped lasso
    off = np.arange(0, G * per + 1, per)
    cl, _ = pds.lin_reg_by(*cols_of(Xg), target=dev(yg), group_offsets=off, l1_reg=0.01, tol=1e-7)
    bo = orc.pl_lr(Xg[:per].astype(np.float64), yg[:per].astype(np.float64), l1_reg=0.01, tol=1e-10, max_iter=2000)
    assert nrel(cl.cpu().numpy()[0], bo) < F32_TOL
    # grouped with more than 64 features, HC3 with more than 16
    pw = 70
    Xw = rng.normal(size=(900, pw)).astype(np.float32)
    yw = (Xw @ rng.normal(size=pw) + 0.1 * rng.normal(size=900)).astype(np.float32)
    cw, nw = pds.lin_reg_by(*cols_of(Xw), target=dev(yw), group_offsets=np.array([0, 400, 900]))
    assert not nw.cpu().numpy().any()
    assert nrel(cw.cpu().numpy()[1], np.linalg.lstsq(Xw[400:].astype(np.float64), yw[400:].astype(np.float64), rcond=None)[0]) < F32_TOL
    r = pds.lin_reg_report(*cols_of(Xw[:, :20]), target=dev(yw), std_err="hc3")
    ro = orc.lin_reg_report(Xw[:, :20].astype(np.float64), yw.astype(np.float64), std_err="hc3")
    assert frel(r["hc3_se"], ro["std_err"], 1e-9) < F32_TOL


# ------------------------------------------------------------------------------------------ f32 twin
def test_f32_path(pds, orc, f32):
    rng = np.random.default_rng(21)
    X, y, _ = make_xy(rng, 400_000, 8, noise=0.05)
    X32, y32 = X.astype(np.float32), y.astype(np.float32)
    truth = orc.pl_lr(X32.astype(np.float64), y32.astype(np.float64), add_bias=True)
    b = pds.lin_reg(*cols_of(X32), target=dev(y32), add_bias=True)
    assert b.dtype == np.float32 and nrel(b, truth) < F32_TOL
    bo = orc.pl_lr(X32, y32, add_bias=True, singular_x_tol=1e-6)  # the reference's all-f32 arithmetic
    assert nrel(b, truth) <= nrel(bo, truth) * 1.5 + 1e-6  # never worse than the f32 reference vs the f64 truth
    b = pds.lin_reg(*cols_of(X32), target=dev(y32), l1_reg=0.001, l2_reg=0.001, tol=1e-7)
    assert nrel(b, orc.pl_lr(X32.astype(np.float64), y32.astype(np.float64), l1_reg=0.001, l2_reg=0.001, tol=1e-9, max_iter=2000)) < F32_TOL
    r = pds.lin_reg_report(*cols_of(X32), target=dev(y32), add_bias=True)
    ro = orc.lin_reg_report(np.c_[X32.astype(np.float64), np.ones(len(y))], y32.astype(np.float64))
    assert frel(r["std_err"], ro["std_err"], 1e-9) < F32_TOL
    co, nu = pds.lin_reg_by(*cols_of(X32), target=dev(y32), group_offsets=np.arange(0, 400_001, 1000), add_bias=False)
    co_o, _ = orc.grouped_lr([y32.astype(np.float64)] + [X32[:, j].astype(np.float64) for j in range(8)], np.arange(0, 400_001, 1000))
    assert np.max(np.linalg.norm(co.cpu().numpy() - co_o, axis=1) / np.linalg.norm(co_o, axis=1)) < F32_TOL
    co, pr, va = pds.rolling_lin_reg(*cols_of(X32[:50_000, :3]), target=dev(y32[:50_000]), window_size=64)
    ref = orc.rolling_lr(X32[:50_000, :3].astype(np.float64), y32[:50_000].astype(np.float64), 64)
    assert np.max(np.linalg.norm(co.cpu().numpy()[63:] - ref, axis=1) / np.linalg.norm(ref, axis=1)) < F32_TOL


@pytest.mark.parametrize("p,bias", [(17, True), (30, False), (31, True), (47, True), (63, True), (64, False)])
def test_grouped_17_to_64_features(pds, orc, p, bias):
    # one wave per group over ceil((p+2)/16) column blocks (grouped_moments_mid_kernel) + the LDS solve
    rng = np.random.default_rng(700 + p)
    G = 300
    sizes = rng.integers(1, 5 * p, size=G)
    sizes[::41] = rng.integers(0, p, size=len(sizes[::41]))  # too-small groups -> null
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    X = rng.normal(size=(N, p))
    y = np.empty(N)
    for g in range(G):
        sl = slice(off[g], off[g + 1])
        y[sl] = X[sl] @ rng.normal(size=p) + 0.1 * rng.normal(size=sizes[g]) + (0.7 if bias else 0.0)
    co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias)
    co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
    co_o, nu_o = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=bias, nthreads=4)
    assert np.array_equal(nu, nu_o) and nu.sum() >= 5
    ok = ~nu
    err = np.linalg.norm(co[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)
    well = sizes[ok] >= 2 * (p + bias) + 8
    assert well.sum() > 50 and np.max(err[well]) < F64_TOL
    co2, nu2 = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias, l2_reg=0.2)
    co2_o, _ = orc.grouped_lr([y] + [X[:, j] for j in range(p)], off, add_bias=bias, l2_reg=0.2, nthreads=4)
    big = sizes >= 2 * (p + bias) + 8  # (tiny groups under a small ridge are as ill-conditioned as without it)
    assert np.max(np.linalg.norm(co2.cpu().numpy()[big] - co2_o[big], axis=1) / np.linalg.norm(co2_o[big], axis=1)) < F64_TOL


@pytest.mark.parametrize("p,bias", [(3, False), (5, True), (8, False), (12, True), (20, True)])
def test_grouped_f32_ragged(pds, orc, f32, p, bias):
    # f32 frames, group boundaries at arbitrary rows (p <= 8 runs the two-slab matrix-core path and its masked steps)
    rng = np.random.default_rng(300 + p)
    G = 2000
    sizes = rng.integers(40, 400, size=G)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(off[-1])
    X = rng.normal(size=(N, p)).astype(np.float32)
    y = np.empty(N, dtype=np.float32)
    for g in range(G):
        sl = slice(off[g], off[g + 1])
        y[sl] = X[sl] @ rng.normal(size=p) + 0.1 * rng.normal(size=sizes[g]) + (0.5 if bias else 0.0)
    co, nu = pds.lin_reg_by(*cols_of(X), target=dev(y), group_offsets=off, add_bias=bias)
    co, nu = co.cpu().numpy(), nu.cpu().numpy().astype(bool)
    assert co.dtype == np.float32 and not nu.any()
    co_o, _ = orc.grouped_lr([y.astype(np.float64)] + [X[:, j].astype(np.float64) for j in range(p)], off, add_bias=bias, nthreads=4)
    assert np.max(np.linalg.norm(co - co_o, axis=1) / np.linalg.norm(co_o, axis=1)) < F32_TOL


# ------------------------------------------------------------------------------------------ full BASELINE sizes
def test_full_size_single_ols_noise_free(pds):
    # configs[1] size: 1e8 rows x 16 f64.  y = X beta exactly  =>  coefficients recover beta, r2 = 1
    import torch

    n, p = 100_000_000, 16
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    xs = [torch.rand(n, dtype=torch.float64, device="cuda", generator=g) for _ in range(p)]
    beta = np.array([(-1.0) ** j * (0.05 + 0.03 * j) for j in range(p)])
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    for j in range(p):
        y.add_(xs[j], alpha=float(beta[j]))
    b = pds.lin_reg(*xs, target=y, add_bias=True)
    assert np.max(np.abs(b[:p] - beta)) < 1e-11 and abs(b[p]) < 1e-11
    # linearity of the Gram build: moments of the frame == sum of moments of two row halves
    h = n // 2 + 12345
    full = pds.gram_moments(*xs, target=y)
    m1 = pds.gram_moments(*[x[:h] for x in xs], target=y[:h])
    m2 = pds.gram_moments(*[x[h:] for x in xs], target=y[h:])
    assert nrel(m1 + m2, full) < 1e-14
    r = pds.lin_reg_report(*xs, target=y, add_bias=False)  # dof = 1e8 - 16: the reference's ppf would hang here
    assert np.all(np.isfinite(r["0.975"])) and abs(r["r2"][0] - 1.0) < 1e-12


def test_full_size_grouped_noise_free(pds):
    # configs[2]-scale: 1e6 groups x 100 rows x 8 feats; y = X beta_g exactly => per-group beta recovered
    import torch

    G, R, p = 1_000_000, 100, 8
    n = G * R
    g = torch.Generator(device="cuda")
    g.manual_seed(2)
    xs = [torch.randn(n, dtype=torch.float64, device="cuda", generator=g) for _ in range(p)]
    bg = torch.randn(G, p, dtype=torch.float64, device="cuda", generator=g)
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    for j in range(p):
        y.add_(xs[j] * bg[:, j].repeat_interleave(R))
    off = torch.arange(0, n + 1, R, dtype=torch.int64, device="cuda")
    co, nu = pds.lin_reg_by(*xs, target=y, group_offsets=off)
    assert int(nu.sum().item()) == 0
    err = (co - bg).norm(dim=1) / bg.norm(dim=1)
    assert float(err.max().item()) < 1e-11
    # with an intercept (centred form): y + group-specific constant => same slopes, the constant as the last coefficient
    cg = torch.randn(G, dtype=torch.float64, device="cuda", generator=g)
    y.add_(cg.repeat_interleave(R))
    co, nu = pds.lin_reg_by(*xs, target=y, group_offsets=off, add_bias=True)
    assert int(nu.sum().item()) == 0 and co.shape[1] == p + 1
    assert float(((co[:, :p] - bg).norm(dim=1) / bg.norm(dim=1)).max().item()) < 1e-10
    assert float((co[:, p] - cg).abs().max().item()) < 1e-10
    # ragged groups on the same frame: sizes 1 .. 300 (mean ~ 100), slopes of a group are still exact
    gen = np.random.default_rng(3)
    sizes = gen.integers(1, 300, size=G)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    offs = offs[offs <= n]
    co, nu = pds.lin_reg_by(*xs, target=y, group_offsets=torch.as_tensor(offs, device="cuda"), add_bias=False, singular_x_tol=1e-12)
    assert co.shape[0] == len(offs) - 1 and bool(torch.isfinite(co[~nu.bool()]).all())


# ------------------------------------------------------------------------------------------ p > 16: tiled-SYRK Gram build
@pytest.mark.parametrize("n,p", [(5000, 17), (20_011, 40), (3000, 130), (40_000, 64), (2500, 126), (2600, 128), (2100, 140),
                                 (2200, 144), (1500, 256)])
def test_wide_moments_and_ols_f64(pds, orc, n, p):
    rng = np.random.default_rng(p)
    X = rng.normal(size=(n, p))
    y = X @ rng.normal(size=p) + 0.3 + 0.1 * rng.normal(size=n)
    M = pds.gram_moments(*cols_of(X), target=dev(y))
    Z = np.c_[X, np.ones(n), y]
    assert nrel(M, Z.T @ Z) < 1e-13
    if p + 1 <= 64:
        b = pds.lin_reg(*cols_of(X), target=dev(y), add_bias=True)
        assert nrel(b, orc.pl_lr(X, y, add_bias=True)) < F64_TOL
    b = pds.lin_reg(*cols_of(X), target=dev(y), add_bias=True, l1_reg=0.01, l2_reg=0.01, tol=1e-9, max_iter=3000)
    assert nrel(b, orc.pl_lr(X, y, add_bias=True, l1_reg=0.01, l2_reg=0.01, tol=1e-9, max_iter=3000)) < 1e-9


@pytest.mark.parametrize("p,bias", [(64, True), (65, False), (100, True), (257, True)])
def test_ols_more_than_64_coefficients(pds, orc, p, bias):
    # p' > 64: Cholesky on an HBM/L2 workspace (solve_big.hip); every solver string, ridge, report (SE), pred, the gate
    rng = np.random.default_rng(500 + p)
    n = 6 * p + 2000
    X = rng.normal(size=(n, p))
    X[:, 1] = 0.5 * X[:, 0] + 0.8 * X[:, 1]
    y = X @ rng.normal(size=p) + 0.25 + 0.1 * rng.normal(size=n)
    for kw in ({}, {"solver": "svd"}, {"solver": "choleskey"}, {"l2_reg": 0.3}):
        b = pds.lin_reg(*cols_of(X), target=dev(y), add_bias=bias, **kw)
        assert nrel(b, orc.pl_lr(X, y, add_bias=bias, **kw)) < F64_TOL
    Xb = np.c_[X, np.ones(n)] if bias else X
    r = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=bias)
    ro = orc.lin_reg_report(Xb, y)
    assert nrel(r["beta"], ro["beta"]) < F64_TOL and frel(r["std_err"], ro["std_err"], 1e-12) < 1e-9
    assert frel(r["t"], ro["t"], 1e-9) < 1e-8 and abs(np.ravel(r["r2"])[0] - np.ravel(ro["r2"])[0]) < 1e-10
    pred, resid = pds.lin_reg(*cols_of(X), target=dev(y), add_bias=bias, return_pred=True)
    bo = orc.pl_lr(X, y, add_bias=bias)
    assert np.max(np.abs(pred.cpu().numpy() - Xb @ bo)) < 1e-9
    Xc = X.copy()
    Xc[:, 7] = Xc[:, 3] - 2.0 * Xc[:, 5]  # exactly collinear: the gate answers null, as the reference does
    assert pds.lin_reg(*cols_of(Xc), target=dev(y), add_bias=bias) is None
    assert orc.pl_lr(Xc, y, add_bias=bias) is None


def test_ols_more_than_64_coefficients_f32(pds, orc, f32):
    rng = np.random.default_rng(77)
    n, p = 20_000, 130
    X = rng.normal(size=(n, p)).astype(np.float32)
    y = (X @ rng.normal(size=p) + 0.1 * rng.normal(size=n)).astype(np.float32)
    b = pds.lin_reg(*cols_of(X), target=dev(y), add_bias=True)
    assert b.dtype == np.float32
    assert nrel(b, orc.pl_lr(X.astype(np.float64), y.astype(np.float64), add_bias=True)) < F32_TOL


@pytest.mark.parametrize("n,p", [(9000, 30), (70_000, 126), (33_333, 128), (10_000, 150), (20_000, 158), (12_345, 256)])
def test_wide_moments_f32_split_k(pds, f32, n, p):
    # f32: <= 8192-row splits in f32 matrix-core tiles, cross-split sum in f64; the [1 | y] tail column block runs narrow
    rng = np.random.default_rng(n + p)
    X = rng.normal(size=(n, p)).astype(np.float32)
    y = rng.normal(size=n).astype(np.float32)
    M = pds.gram_moments(*cols_of(X), target=dev(y))
    Z = np.c_[X.astype(np.float64), np.ones(n), y.astype(np.float64)]
    G = Z.T @ Z
    assert M.dtype == np.float32 and M.shape == G.shape
    assert np.max(np.abs(M - G) / np.sqrt(np.outer(np.diag(G), np.diag(G)))) < 2e-6
    assert np.array_equal(M, M.T)


@pytest.mark.parametrize("n,p,weighted", [(20_011, 515, False), (9_001, 260, True), (40_000, 768, False), (16_390, 512, True)])
def test_wide_moments_f32_256_tile(pds, f32, n, p, weighted):
    """Even block grids (p = 255..286, 511..542, 767..798) take the 256 x 256 tile of the bf16-split arithmetic: diagonal and
    off-diagonal workgroups, the fused [1 | y] tail, ragged last stage, the weight column -- against an f64 Gram, with both f32
    arithmetics on the same frame."""
    import os

    rng = np.random.default_rng(n + p)
    X = (rng.normal(size=(n, p)) + rng.normal(size=p)).astype(np.float32)
    y = rng.normal(size=n).astype(np.float32)
    w = (rng.random(n) + 0.25).astype(np.float32) if weighted else None
    Z = np.c_[X.astype(np.float64), np.ones(n), y.astype(np.float64)]
    G = Z.T @ (Z if w is None else w.astype(np.float64)[:, None] * Z)
    sc = np.sqrt(np.outer(np.diag(G), np.diag(G)))
    got = {}
    for name, native in (("tile256", 0), ("f32", 1)):
        pds.default_context().set_option("wide_f32_native", native)
        try:
            M = pds.gram_moments(*cols_of(X), target=dev(y), weights=None if w is None else dev(w))
        finally:
            pds.default_context().set_option("wide_f32_native", 0)
        assert M.dtype == np.float32 and M.shape == G.shape and np.array_equal(M, M.T)
        assert np.max(np.abs(M - G) / sc) < 3e-6, name
        got[name] = M
    # (the 128 x 128 tile of the split arithmetic serves the odd block grids -- test_wide_moments_* -- and adds the same products in
    #  the same order; forcing it onto these grids is a development switch, PDS_WIDE_TILE128 with EXTRA=-DPDS_DEV_SWITCHES)


def test_config5_elastic_net_f32_wide(pds, orc, f32):
    # configs[4] at reduced N: elastic net, p = 512 f32, AR(0.5)-correlated columns, 32 non-zero coefficients
    rng = np.random.default_rng(4)
    n, p = 60_000, 512
    E = rng.normal(size=(n, p)).astype(np.float32)
    X = np.empty_like(E)
    X[:, 0] = E[:, 0]
    for j in range(1, p):
        X[:, j] = 0.5 * X[:, j - 1] + np.sqrt(0.75) * E[:, j]
    beta = np.zeros(p)
    beta[rng.choice(p, 32, replace=False)] = rng.normal(size=32)
    y = (X @ beta + 0.5 * rng.normal(size=n)).astype(np.float32)
    import os

    Z = np.c_[X.astype(np.float64), np.ones(n), y.astype(np.float64)]
    G = Z.T @ Z
    # the f32 Gram beyond 16 features has two arithmetics (moments_wide.hip): products on the bf16 matrix cores as exact
    # three-plane splits (default; measured 5.1e-7 here) or v_mfma_f32_32x32x2_f32 (context option "wide_f32_native"; 1.3e-7)
    for native, bar in (("1", 3e-7), ("0", 1e-6)):
        pds.default_context().set_option("wide_f32_native", int(native))
        try:
            M = pds.gram_moments(*cols_of(X), target=dev(y))
        finally:
            pds.default_context().set_option("wide_f32_native", 0)
        assert nrel(M, G) < bar, (native, nrel(M, G))
        assert np.array_equal(M, M.T)
    b = pds.lin_reg(*cols_of(X), target=dev(y), l1_reg=0.01, l2_reg=0.01, tol=1e-5)
    truth = orc.pl_lr(X.astype(np.float64), y.astype(np.float64), l1_reg=0.01, l2_reg=0.01, tol=1e-9, max_iter=2000)
    o32 = orc.pl_lr(X, y, l1_reg=0.01, l2_reg=0.01, tol=1e-5, max_iter=2000)  # the reference's all-f32 sweeps, same stopping rule
    assert b.dtype == np.float32
    hold_f32("config 5 at 6e4 x 512", b, o32, truth)  # (the configured 1e7 / 1e6 rows: tests/test_baseline_sizes.py)
    assert np.sum(np.abs(b) > 1e-6) < 200  # sparse solution


# ------------------------------------------------------------------------------------------ null policies (Arrow validity)
def _arrow_frame(rng, n, p, null_cols, frac=0.07, offset=0):
    import pyarrow as pa

    X = rng.random((n, p))
    y = X @ rng.normal(size=p) + 0.2 + 0.05 * rng.normal(size=n)
    masks = {}
    arrs = []
    for c, v in enumerate([y] + [X[:, j] for j in range(p)]):
        m = (rng.random(n) < frac) if c in null_cols else np.zeros(n, dtype=bool)
        masks[c] = m
        a = pa.array(np.where(m, 123456.0, v), mask=m)  # null slots hold junk on purpose
        arrs.append(a.slice(offset) if offset else a)
    valid = [~masks[c][offset:] for c in range(p + 1)]
    return arrs, X[offset:], y[offset:], valid


@pytest.mark.parametrize("offset", [0, 5])
def test_null_policies_match_reference_semantics(pds, orc, offset):
    # tests/test_many.py:1636-1726 + tests/test_linear_exprs.py:411-432: skip / raise / zero / one / "0.5" / ignore / invalid
    from polars_ds_extension_amd._lib import PdsError

    rng = np.random.default_rng(7)
    n, p = 40_000, 3
    arrs, X, y, valid = _arrow_frame(rng, n, p, null_cols={1, 3}, offset=offset)
    keep = np.logical_and.reduce(valid)
    b = pds.lin_reg(*arrs[1:], target=arrs[0], add_bias=True, null_policy="skip")
    assert nrel(b, orc.pl_lr(X[keep], y[keep], add_bias=True)) < F64_TOL
    pred, resid, rv = pds.lin_reg(*arrs[1:], target=arrs[0], add_bias=True, null_policy="skip", return_pred=True)
    assert np.array_equal(rv, keep) and np.isnan(pred[~keep]).all()
    ref = np.c_[X[keep], np.ones(keep.sum())] @ b
    assert nrel(pred[keep], ref) < F64_TOL and np.max(np.abs(resid[keep] - (y[keep] - ref))) < 1e-11
    for pol, fill in (("zero", 0.0), ("one", 1.0), ("0.5", 0.5), ("ZERO", 0.0)):
        Xf = X.copy()
        for j in range(p):
            Xf[~valid[j + 1], j] = fill
        b = pds.lin_reg(*arrs[1:], target=arrs[0], add_bias=True, null_policy=pol)
        assert nrel(b, orc.pl_lr(Xf, y, add_bias=True)) < F64_TOL  # the target has no nulls here: nothing is dropped
    with pytest.raises(PdsError, match="Nulls found in data"):
        pds.lin_reg(*arrs[1:], target=arrs[0], null_policy="raise")
    with pytest.raises(ValueError, match="Invalid NullPolicy"):
        pds.lin_reg(*arrs[1:], target=arrs[0], null_policy="nonsense")
    b = pds.lin_reg(*arrs[1:], target=arrs[0], null_policy="ignore", singular_x_tol=0.0)
    assert np.isnan(b).all()  # nulls become NaN and poison the normal equations, as in the reference
    # target nulls + fill: rows with a null target are dropped, feature nulls filled (linear_regression.rs:207-227)
    arrs2, X2, y2, valid2 = _arrow_frame(rng, n, p, null_cols={0, 2}, offset=offset)
    Xf = X2.copy()
    Xf[~valid2[2], 1] = 0.0
    k2 = valid2[0]
    b = pds.lin_reg(*arrs2[1:], target=arrs2[0], null_policy="zero")
    assert nrel(b, orc.pl_lr(Xf[k2], y2[k2])) < F64_TOL
    # a frame without nulls takes the fast path under every policy, incl. raise
    arrs3, X3, y3, _ = _arrow_frame(rng, 5000, p, null_cols=set())
    assert nrel(pds.lin_reg(*arrs3[1:], target=arrs3[0], null_policy="raise"), orc.pl_lr(X3, y3)) < F64_TOL


@pytest.mark.parametrize("se", ["se", "hc1"])
def test_report_null_policies(pds, orc, se):
    # pl_lin_reg_report runs series_to_mat_for_lr with the expression's null_policy (linear_regression.rs:846)
    from polars_ds_extension_amd._lib import PdsError

    rng = np.random.default_rng(17)
    n, p = 30_000, 4
    arrs, X, y, valid = _arrow_frame(rng, n, p, null_cols={0, 2, 4}, offset=3)
    keep = np.logical_and.reduce(valid)
    yv = float(np.var(y[valid[0]], ddof=1))
    key = {"se": "std_err"}.get(se, f"{se}_se")
    r = pds.lin_reg_report(*arrs[1:], target=arrs[0], add_bias=True, std_err=se, null_policy="skip", y_var=yv)
    ro = orc.lin_reg_report(np.c_[X[keep], np.ones(keep.sum())], y[keep], y_var=yv, std_err=se)
    assert nrel(r["beta"], ro["beta"]) < F64_TOL and frel(r[key], ro["std_err"], 1e-12) < 1e-9
    assert frel(r["p>|t|"], ro["p"], 1e-12) < 1e-7 and abs(r["r2"][0] - ro["r2"]) < 1e-12 and abs(r["adj_r2"][0] - ro["adj_r2"]) < 1e-12
    r2 = pds.lin_reg_report(*arrs[1:], target=arrs[0], add_bias=True, std_err=se, null_policy="skip")  # y_var from pyarrow
    assert abs(r2["r2"][0] - r["r2"][0]) < 1e-12
    Xf = X.copy()
    for j in range(p):
        Xf[~valid[j + 1], j] = 1.0
    k0 = valid[0]
    r = pds.lin_reg_report(*arrs[1:], target=arrs[0], add_bias=True, std_err=se, null_policy="one", y_var=yv)
    ro = orc.lin_reg_report(np.c_[Xf[k0], np.ones(k0.sum())], y[k0], y_var=yv, std_err=se)
    assert nrel(r["beta"], ro["beta"]) < F64_TOL and frel(r[key], ro["std_err"], 1e-12) < 1e-9
    with pytest.raises(PdsError, match="Nulls found in data"):
        pds.lin_reg_report(*arrs[1:], target=arrs[0], add_bias=True, null_policy="raise", y_var=yv)


def test_grouped_null_policies(pds, orc):
    # group_by(key).agg(pds.lin_reg(null_policy=...)): every group is fitted on its rows that survive the policy
    from polars_ds_extension_amd._lib import PdsError

    rng = np.random.default_rng(23)
    p = 3
    G = 400
    sizes = rng.integers(1, 80, size=G)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(off[-1])
    arrs, X, y, valid = _arrow_frame(rng, n, p, null_cols={0, 2}, frac=0.15)
    keep = np.logical_and.reduce(valid)
    co, nu = pds.lin_reg_by(*arrs[1:], target=arrs[0], group_offsets=off, add_bias=True, null_policy="skip")
    rank = np.concatenate([[0], np.cumsum(keep)])
    off_k = rank[off]
    co_o, nu_o = orc.grouped_lr([y[keep]] + [X[keep][:, j] for j in range(p)], off_k, add_bias=True, nthreads=2)
    assert np.array_equal(nu.astype(bool), nu_o) and nu_o.sum() > 3
    ok = ~nu_o & (np.diff(off_k) >= 2 * (p + 1) + 8)
    assert np.max(np.linalg.norm(co[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)) < F64_TOL
    # fill: feature nulls become the fill value, rows with a null target are dropped
    Xf = X.copy()
    Xf[~valid[2], 1] = 0.0
    k0 = valid[0]
    rank0 = np.concatenate([[0], np.cumsum(k0)])
    co, nu = pds.lin_reg_by(*arrs[1:], target=arrs[0], group_offsets=off, add_bias=True, null_policy="zero")
    co_o, nu_o = orc.grouped_lr([y[k0]] + [Xf[k0][:, j] for j in range(p)], rank0[off], add_bias=True, nthreads=2)
    assert np.array_equal(nu.astype(bool), nu_o)
    ok = ~nu_o & (np.diff(rank0[off]) >= 2 * (p + 1) + 8)
    assert np.max(np.linalg.norm(co[ok] - co_o[ok], axis=1) / np.linalg.norm(co_o[ok], axis=1)) < F64_TOL
    with pytest.raises(PdsError, match="Nulls found in data"):
        pds.lin_reg_by(*arrs[1:], target=arrs[0], group_offsets=off, null_policy="raise")


def test_linear_impute_frame(pds):
    # tests/test_transforms.py:33-48 (`linear_impute`: pipeline/transforms.py:115-155 = lin_reg(null_policy="skip") + fill_null):
    # c = a + b with one null; the skip-null fit must return beta = [1, 1] and the null must be imputed as 6.0
    import pyarrow as pa

    a = pa.array([3.0, 2, 3, 4, 5, 6, 7, 8, 9, 11])
    b = pa.array([1.0, 2, 3, 4, 5, 6, 7, 8, 9, 10])
    c = pa.array([4.0, 4.0, None, 8.0, 10.0, 12.0, 14.0, 16.0, 18.0, 21.0])
    beta = pds.lin_reg(a, b, target=c, add_bias=False, null_policy="skip")
    np.testing.assert_allclose(beta, [1.0, 1.0], rtol=0, atol=1e-10)
    filled = np.where(np.asarray(c.is_null()), np.asarray(a) * beta[0] + np.asarray(b) * beta[1], np.asarray(c.fill_null(0.0)))
    np.testing.assert_allclose(filled, np.asarray(a) + np.asarray(b), atol=1e-9)


def test_literal_skip_null_frame(pds):
    # tests/test_linear_exprs.py:411-432 (literal frame): a null in row 0 -> pred [None, 9.5, 10.5, 11.5, 12.5], resid [None, 0, 0, 0, 0]
    import pyarrow as pa

    x = pa.array([None, 2.0, 3.0, 4.0, 5.0])
    y = pa.array([8.5, 9.5, 10.5, 11.5, 12.5])
    pred, resid, valid = pds.lin_reg(x, target=y, add_bias=True, null_policy="skip", return_pred=True)
    assert list(valid) == [False, True, True, True, True]
    np.testing.assert_allclose(pred[1:], [9.5, 10.5, 11.5, 12.5], atol=1e-10)
    np.testing.assert_allclose(resid[1:], 0.0, atol=1e-10)


# ------------------------------------------------------------------------------------------ multi-target
@pytest.mark.parametrize("p", [5, 15, 16, 22, 70])
def test_multi_target(pds, orc, p):
    # tests/test_linear_exprs.py:1069-1113 (struct fields target_i == single-target fits, 1e-12) and :376-408 (pred)
    rng = np.random.default_rng(4)
    n = 30_000
    X = rng.normal(size=(n, p))
    Y = np.c_[X @ rng.normal(size=p) + 0.3, X @ rng.normal(size=p) - 1.0, rng.normal(size=n)] + 0.05 * rng.normal(size=(n, 3))
    for bias, lam in ((False, 0.0), (True, 0.0), (True, 0.2)):
        out = pds.lin_reg(*cols_of(X), target=[dev(Y[:, i]) for i in range(3)], add_bias=bias, l2_reg=lam)
        assert list(out) == ["target_0", "target_1", "target_2"]
        for i in range(3):
            single = pds.lin_reg(*cols_of(X), target=dev(Y[:, i]), add_bias=bias, l2_reg=lam)
            assert nrel(out[f"target_{i}"], single) < 1e-11
            assert nrel(out[f"target_{i}"], orc.pl_lr(X, Y[:, i], add_bias=bias, l2_reg=lam)) < F64_TOL
    pr = pds.lin_reg(*cols_of(X), target=[dev(Y[:, 0]), dev(Y[:, 1])], add_bias=True, return_pred=True)
    Xb = np.c_[X, np.ones(n)]
    for i in range(2):
        b = orc.pl_lr(X, Y[:, i], add_bias=True)
        assert nrel(pr[f"target_{i}_pred"].cpu().numpy(), Xb @ b) < F64_TOL
        assert np.max(np.abs(pr[f"target_{i}_resid"].cpu().numpy() - (Y[:, i] - Xb @ b))) < 1e-10
    Xc = np.c_[X[:, 0], 2 * X[:, 0], X[:, 1]]
    out = pds.lin_reg(*cols_of(Xc), target=[dev(Y[:, 0]), dev(Y[:, 1])])
    assert out["target_0"] is None and out["target_1"] is None  # the gate depends on X only: all targets null


@pytest.mark.parametrize("p,bias", [(17, True), (32, True), (33, False), (40, False), (64, True), (130, True)])
def test_wide_weighted_and_hc(pds, orc, p, bias):
    # p > 16 with a weight column: X' diag(w) X as the Gram of sqrt(w) Z; WLS report; HC0 / HC1 sandwich
    rng = np.random.default_rng(900 + p)
    n = 12_000 + 50 * p
    X = rng.normal(size=(n, p))
    y = X @ rng.normal(size=p) + 0.4 + rng.normal(size=n) * (0.5 + np.abs(X[:, 0]))
    w = rng.random(n) + 0.25
    Xb = np.c_[X, np.ones(n)] if bias else X
    b = pds.lin_reg(*cols_of(X), target=dev(y), add_bias=bias, weights=dev(w))
    assert nrel(b, orc.pl_lr(X, y, add_bias=bias, weights=w)) < F64_TOL
    M = pds.gram_moments(*cols_of(X), target=dev(y), weights=dev(w))
    Z = np.c_[X, np.ones(n), y]
    assert nrel(M, Z.T @ (w[:, None] * Z)) < 1e-13
    r = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=bias, weights=dev(w))
    ro = orc.wls_report(Xb, y, w)
    assert nrel(r["beta"], ro["beta"]) < F64_TOL and frel(r["std_err"], ro["std_err"], 1e-12) < 1e-9
    for se in ("hc0", "hc1", "hc2", "hc3"):  # hc2 / hc3: per-row leverages x_i' (X'X)^-1 x_i on the device
        r = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=bias, std_err=se)
        ro = orc.lin_reg_report(Xb, y, std_err=se)
        assert nrel(r["beta"], ro["beta"]) < F64_TOL and frel(r[f"{se}_se"], ro["std_err"], 1e-12) < 1e-9
        assert frel(r["t"], ro["t"], 1e-6) < 1e-8 and frel(r["p>|t|"], ro["p"], 1e-12) < 1e-6
        # 17 .. 64 features: residuals, leverages and the meat come out of ONE kernel (moments_mid.hip FUSE) -- sum e^2 too
        assert abs(r["r2"][0] - ro["r2"]) < 1e-11


def test_wide_pred_and_report(pds, orc):
    rng = np.random.default_rng(31)
    n, p = 30_000, 24
    X = rng.normal(size=(n, p))
    beta = rng.normal(size=p)
    beta[[3, 7]] = 0.0
    y = X @ beta + 0.4 + rng.normal(size=n)
    pred, resid = pds.lin_reg(*cols_of(X), target=dev(y), add_bias=True, return_pred=True)
    b = orc.pl_lr(X, y, add_bias=True)
    assert nrel(pred.cpu().numpy(), np.c_[X, np.ones(n)] @ b) < F64_TOL
    r = pds.lin_reg_report(*cols_of(X), target=dev(y), add_bias=True, y_var=float(np.var(y, ddof=1)))
    ro = orc.lin_reg_report(np.c_[X, np.ones(n)], y)
    assert nrel(r["beta"], ro["beta"]) < F64_TOL and frel(r["std_err"], ro["std_err"], 1e-12) < F64_TOL
    assert frel(r["p>|t|"], ro["p"], 1e-12) < 1e-8 and abs(r["r2"][0] - ro["r2"]) < 1e-12


def _ar_design(x, lag, bias):
    n = len(x)
    A = np.column_stack([x[lag - i : n - i] for i in range(1, lag + 1)] + ([np.ones(n - lag)] if bias else []))
    return A, x[lag:]


@pytest.mark.parametrize("lag,bias", [(1, True), (3, True), (5, False), (16, True)])
def test_query_ar_coeffs(pds, lag, bias):
    """exprs/ts_features.py:419-461 (SURVEY 8f rank 4 consumer): lin_reg on lagged views of ONE series -- the feature columns
    are pointers 8 bytes apart into the same buffer, on the host and in HBM."""
    rng = np.random.default_rng(40 + lag)
    n = 200_003
    phi = 0.5 ** np.arange(1, lag + 1) * (-1.0) ** np.arange(lag)
    from scipy.signal import lfilter

    x = lfilter([1.0], np.concatenate([[1.0], -phi]), 0.3 + rng.normal(size=n))  # AR(lag) process
    A, b = _ar_design(x, lag, bias)
    ref = np.linalg.lstsq(A, b, rcond=None)[0]
    assert nrel(pds.query_ar_coeffs(x, lag, add_bias=bias), ref) < F64_TOL
    assert nrel(pds.query_ar_coeffs(dev(x), lag, add_bias=bias), ref) < F64_TOL
    with pytest.raises(ValueError, match="lag"):
        pds.query_ar_coeffs(x, 0)
    with pytest.raises(ValueError, match="null_polocy"):
        pds.query_ar_coeffs(x, 2, null_policy="skip")


def test_query_ar_coeffs_arrow_nulls(pds):
    import pyarrow as pa

    rng = np.random.default_rng(9)
    n, lag = 50_000, 4
    x = np.cumsum(rng.normal(size=n)) * 0.01 + rng.normal(size=n)
    mask = rng.random(n) < 0.01
    mask[:lag] = False
    xa = pa.array(x, mask=mask)
    with pytest.raises(Exception, match="Nulls found"):
        pds.query_ar_coeffs(xa, lag)
    # "zero": null features become 0, rows whose target is null are dropped (series_to_mat_for_lr :211-240)
    xz = np.where(mask, 0.0, x)
    A, b = _ar_design(xz, lag, True)
    keep = ~mask[lag:]
    ref = np.linalg.lstsq(A[keep], b[keep], rcond=None)[0]
    assert nrel(pds.query_ar_coeffs(xa, lag, null_policy="zero"), ref) < F64_TOL


def test_device_columns_at_element_alignment(pds, orc):
    """Arrow slices put column starts at any multiple of the element size: every kernel takes 8-byte (f64) / 4-byte (f32)
    aligned device pointers.  Results equal those on 256-byte aligned clones of the same data."""
    import torch

    rng = np.random.default_rng(123)
    n, p = 70_001, 5
    X, y, _ = make_xy(rng, n, p)
    big = [torch.from_numpy(np.concatenate([np.zeros(j + 1), X[:, j]])).cuda() for j in range(p)]
    xs_u = [b[j + 1 :] for j, b in enumerate(big)]           # starts 8, 16, 24 ... bytes past an allocation boundary
    yb = torch.from_numpy(np.concatenate([np.zeros(3), y])).cuda()
    y_u = yb[3:]
    xs_a, y_a = cols_of(X), dev(y)
    assert any(t.data_ptr() % 16 for t in xs_u) and y_u.data_ptr() % 16 == 8
    for bias in (False, True):
        assert nrel(pds.lin_reg(*xs_u, target=y_u, add_bias=bias), pds.lin_reg(*xs_a, target=y_a, add_bias=bias)) < 1e-13
        pu, ru = pds.lin_reg(*xs_u, target=y_u, add_bias=bias, return_pred=True)
        pa_, ra = pds.lin_reg(*xs_a, target=y_a, add_bias=bias, return_pred=True)
        assert nrel(pu.cpu().numpy(), pa_.cpu().numpy()) < 1e-13 and nrel(ru.cpu().numpy(), ra.cpu().numpy()) < 1e-9
    ru = pds.lin_reg_report(*xs_u, target=y_u, add_bias=True, std_err="hc1")
    ra = pds.lin_reg_report(*xs_a, target=y_a, add_bias=True, std_err="hc1")
    for k in ("beta", "hc1_se", "t"):
        assert nrel(ru[k], ra[k]) < 1e-10
    cu, pu, vu = pds.rolling_lin_reg(*xs_u, target=y_u, window_size=64, add_bias=True)
    ca, pa2, va = pds.rolling_lin_reg(*xs_a, target=y_a, window_size=64, add_bias=True)
    assert torch.equal(vu, va) and nrel(cu[63:].cpu().numpy(), ca[63:].cpu().numpy()) < 1e-12
    off = np.arange(0, n, 97, dtype=np.int64)
    off = np.concatenate([off, [n]]) if off[-1] != n else off
    cu, nu = pds.lin_reg_by(*xs_u, target=y_u, group_offsets=off)
    ca, na = pds.lin_reg_by(*xs_a, target=y_a, group_offsets=off)
    assert torch.equal(nu, na) and nrel(cu.cpu().numpy(), ca.cpu().numpy()) < 1e-12
    # f32: 4-byte aligned starts
    pds.config.LIN_REG_EXPR_F64 = False
    try:
        big32 = [torch.from_numpy(np.concatenate([np.zeros(j + 1), X[:, j]]).astype(np.float32)).cuda() for j in range(p)]
        xs32 = [b[j + 1 :] for j, b in enumerate(big32)]
        y32 = torch.from_numpy(np.concatenate([np.zeros(1), y]).astype(np.float32)).cuda()[1:]
        got = pds.lin_reg(*xs32, target=y32)
        want = pds.lin_reg(*[t.clone() for t in xs32], target=y32.clone())
        assert nrel(got, want) < 1e-6
    finally:
        pds.config.LIN_REG_EXPR_F64 = True
