"""
The reference's own tests of the least-squares path (/root/reference/tests/test_linear_exprs.py), re-stated one by one
against this implementation: same frames (the reference's seeds where it seeds, `default_rng` where it draws from its
unseeded `pds.random`), same oracles (scikit-learn, numpy), same tolerances.  Data goes in as host column buffers
(NumPy = what the Rust host unwraps from Arrow) and through the C ABI to the MI355X.  `group_by` tests use the key-aware
entry point (`lin_reg_by_key`, the `pl_lr_by` symbol).  Each test names the reference test and its lines.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pds():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import polars_ds_extension_amd as m

    m.config.LIN_REG_EXPR_F64 = True
    return m


@pytest.fixture(params=["f64", "f32"])
def lin_reg_dtype(request, pds):
    """`lin_reg_dtype` of the reference (:1184-1188): every gate test runs under the f64 and the f32 symbols."""
    pds.config.LIN_REG_EXPR_F64 = request.param == "f64"
    yield request.param
    pds.config.LIN_REG_EXPR_F64 = True


def cols(X):
    return [np.ascontiguousarray(X[:, j]) for j in range(X.shape[1])]


def frame3(seed, size=5_000, beta=(0.5, 0.25, -0.15), noise=1e-4):
    """pds.frame(size).select(random x1, x2, x3).with_columns(y = ... + random * 1e-4)"""
    rng = np.random.default_rng(seed)
    X = rng.random((size, 3))
    return X, X @ np.asarray(beta) + noise * rng.random(size)


# ------------------------------------------------------------------------------------------------ OLS / ridge
def test_lin_reg_against_sklearn(pds):  # :61-120
    from sklearn import linear_model

    X, y = frame3(61, beta=(0.5, 0.1, -0.15))
    reg = linear_model.LinearRegression(fit_intercept=True).fit(X, y)
    b = pds.lin_reg(*cols(X), target=y, add_bias=True)
    assert np.all(np.abs(b[:3] - reg.coef_) < 1e-5) and np.isclose(b[-1], reg.intercept_, rtol=1e-5)
    reg = linear_model.Ridge(alpha=0.1, fit_intercept=True).fit(X, y)
    b = pds.lin_reg(*cols(X), target=y, l2_reg=0.1, add_bias=True)
    assert np.all(np.abs(b[:3] - reg.coef_) < 1e-3) and abs(b[-1] - reg.intercept_) < 1e-3


def test_lin_reg_single_big_fit_no_regression_path(pds):  # :1116-1142
    from sklearn.linear_model import LinearRegression

    rng = np.random.default_rng(5)
    n, p = 50_000, 6
    X = rng.normal(size=(n, p))
    y = X @ rng.normal(size=p) + 1.5 + rng.normal(size=n) * 0.1
    b = pds.lin_reg(*cols(X), target=y, add_bias=True)
    sk = LinearRegression().fit(X, y)
    np.testing.assert_allclose(b[:-1], sk.coef_, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(b[-1], sk.intercept_, rtol=1e-8, atol=1e-10)


def test_lin_reg_with_bias_appended_column_equivalence(pds):  # :956-981
    rng = np.random.default_rng(1)
    n = 500
    x1, x2, y = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    with_flag = pds.lin_reg(x1, x2, target=y, add_bias=True)
    manual = pds.lin_reg(x1, x2, np.ones(n), target=y, add_bias=False)
    np.testing.assert_allclose(with_flag, manual, rtol=1e-10, atol=1e-12)


def test_lin_reg_many_small_groups_matches_per_group(pds):  # :918-953
    rng = np.random.default_rng(0)
    n_groups, per = 200, 25
    g = np.repeat(np.arange(n_groups), per)
    X = rng.standard_normal((n_groups * per, 2))
    y = X @ [1.5, -0.5] + rng.standard_normal(n_groups * per)
    keys, co, nu = pds.lin_reg_by_key(*cols(X), target=y, key=g)
    assert np.array_equal(keys, np.arange(n_groups)) and not nu.any()
    for k in range(n_groups):
        m = g == k
        np.testing.assert_allclose(co[k], pds.lin_reg(*cols(X[m]), target=y[m]), rtol=1e-12, atol=1e-12)


def test_lin_reg_in_group_by(pds):  # :435-474 (literal frame): group_by(...).agg(lin_reg(return_pred=True)) == filter + fit
    A = np.array([1] * 4 + [2] * 4)
    Y = np.ones(8)
    X1 = np.arange(1.0, 9.0)
    X2 = np.array([2.0, 3, 4, 1, 6, 7, 8, 5])
    keys, co, nu = pds.lin_reg_by_key(X1, X2, target=Y, key=A)
    for i, k in enumerate((1, 2)):
        m = A == k
        b = pds.lin_reg(X1[m], X2[m], target=Y[m], add_bias=False)
        np.testing.assert_allclose(co[i], b, rtol=1e-12, atol=1e-12)
        pred, resid = pds.lin_reg(X1[m], X2[m], target=Y[m], add_bias=False, return_pred=True)
        np.testing.assert_allclose(pred, np.c_[X1[m], X2[m]] @ co[i], atol=1e-12)
        np.testing.assert_allclose(resid, Y[m] - pred, atol=1e-12)


def test_lin_reg_null_skip_in_small_group(pds):  # :1145-1176 (literal frame): nulls + the many-small-groups path
    import pyarrow as pa

    g = np.array([1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3])
    x = [1.0, 2.0, None, 4.0, 1.0, None, 3.0, 4.0, 1.0, 2.0, 3.0, 4.0]
    y = [2.0, 4.0, 6.0, 8.0, 1.0, 2.0, 3.0, None, 0.5, 1.0, 1.5, 2.0]
    co, nu = pds.lin_reg_by(pa.array(x), target=pa.array(y), group_offsets=[0, 4, 8, 12], add_bias=True, null_policy="skip")
    assert not nu.any()
    for i in range(3):
        rows = [j for j in range(4 * i, 4 * i + 4) if x[j] is not None and y[j] is not None]
        exp = pds.lin_reg(np.array([x[j] for j in rows]), target=np.array([y[j] for j in rows]), add_bias=True)
        np.testing.assert_allclose(co[i], exp, rtol=1e-12, atol=1e-12)


def test_lin_reg_skip_null(pds):  # :411-432 (literal frame)
    import pyarrow as pa

    x = pa.array([None, 2.0, 3.0, 4.0, 5.0])
    y = pa.array([8.5, 9.5, 10.5, 11.5, 12.5])
    pred, resid, valid = pds.lin_reg(x, target=y, add_bias=True, null_policy="skip", return_pred=True)
    assert list(valid) == [False, True, True, True, True]
    np.testing.assert_allclose(pred[1:], [9.5, 10.5, 11.5, 12.5], atol=1e-10)
    np.testing.assert_allclose(resid[1:], 0.0, atol=1e-10)


# ------------------------------------------------------------------------------------------------ multi-target, rcond
def test_lin_reg_multi_target_struct_output(pds):  # :1069-1113
    rng = np.random.default_rng(4)
    n = 400
    X = rng.standard_normal((n, 3))
    Y = np.c_[X @ [1.0, 2.0, -1.0] + 0.1 * rng.standard_normal(n), X @ [0.5, 0.0, 3.0] + 0.1 * rng.standard_normal(n)]
    out = pds.lin_reg(*cols(X), target=[Y[:, 0], Y[:, 1]], add_bias=True)
    assert list(out) == ["target_0", "target_1"]
    for i in range(2):
        np.testing.assert_allclose(out[f"target_{i}"], pds.lin_reg(*cols(X), target=Y[:, i], add_bias=True), rtol=1e-12, atol=1e-12)


def test_pl_lr_multi_pred_correctness(pds):  # :376-408
    rng = np.random.default_rng(42)
    n = 300
    X = rng.standard_normal((n, 2))
    Y = np.c_[X @ [1.0, -2.0] + 0.3, X @ [0.2, 0.7] - 1.0] + 0.05 * rng.standard_normal((n, 2))
    out = pds.lin_reg(*cols(X), target=[Y[:, 0], Y[:, 1]], add_bias=True, return_pred=True)
    for i in range(2):
        b = pds.lin_reg(*cols(X), target=Y[:, i], add_bias=True)
        pred = np.c_[X, np.ones(n)] @ b
        np.testing.assert_allclose(out[f"target_{i}_pred"], pred, atol=1e-8)
        np.testing.assert_allclose(out[f"target_{i}_resid"], Y[:, i] - pred, atol=1e-8)


def test_lin_reg_with_rcond(pds):  # :477-512
    X, y = frame3(477, size=5_000, beta=(0.2, 0.3, 0.5), noise=0.1)
    coeffs, sv = pds.lin_reg_w_rcond(*cols(X), target=y, rcond=0.3)
    ref, _, _, sv_ref = np.linalg.lstsq(X, y, rcond=0.3)
    assert np.all(np.abs(coeffs - ref) < 1e-10) and np.all(np.abs(sv - sv_ref) < 1e-10)


def test_lin_reg_with_rcond_truncates_singular_value(pds):  # :515-554
    rng = np.random.default_rng(123)
    n = 2000
    x1 = rng.standard_normal(n)
    x2 = x1 + rng.standard_normal(n) * 1e-6
    x3 = rng.standard_normal(n)
    y = x1 + 0.5 * x2 - 0.3 * x3
    rcond = 1e-3
    coeffs, svs = pds.lin_reg_w_rcond(x1, x2, x3, target=y, rcond=rcond)
    X = np.c_[x1, x2, x3]
    evals, evecs = np.linalg.eigh(X.T @ X)
    thr = rcond * np.sqrt(evals.max())
    assert (evals < thr).any()
    pinv = sum((1.0 / ev) * np.outer(vec, vec) for ev, vec in zip(evals, evecs.T) if ev >= thr)
    np.testing.assert_allclose(coeffs, pinv @ (X.T @ y), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(np.sort(svs)[::-1], np.sqrt(np.sort(evals))[::-1], rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------------------------------------ lasso / elastic net / NNLS
def test_lasso_regression(pds):  # :557-604
    from sklearn import linear_model

    X, y = frame3(557)
    for lam in (0.01, 0.05, 0.1, 0.2):
        sk = linear_model.Lasso(alpha=lam, fit_intercept=False).fit(X, y)
        assert np.all(np.abs(sk.coef_ - pds.lin_reg(*cols(X), target=y, l1_reg=lam, add_bias=False)) < 1e-4)
    for lam in (0.01, 0.05, 0.1, 0.2):
        b = pds.lin_reg(*cols(X), target=y, l1_reg=lam, add_bias=True)
        sk = linear_model.Lasso(alpha=lam, fit_intercept=True).fit(X, y)
        assert np.all(np.abs(sk.coef_ - b[:3]) < 1e-4) and abs(b[-1] - sk.intercept_) < 1e-4


def test_positive_lin_reg(pds):  # :607-674
    from sklearn.linear_model import ElasticNet, LinearRegression

    X, y = frame3(607)
    for bias in (True, False):
        b = pds.lin_reg(*cols(X), target=y, positive=True, add_bias=bias)
        assert np.all((b[:-1] if bias else b) >= 0.0)
        sk = LinearRegression(positive=True, fit_intercept=bias).fit(X, y)
        assert np.all(np.isclose(b[:3], sk.coef_, atol=1e-5))
        if bias:
            assert np.isclose(float(b[-1]), sk.intercept_, atol=1e-5)
    for reg, bias in zip((0.01, 0.05, 0.1, 0.2), (False, True, False, True)):
        b = pds.lin_reg(*cols(X), target=y, l1_reg=reg, l2_reg=reg, add_bias=bias)
        assert np.all((b[:-1] if bias else b) >= 0.0)
        sk = ElasticNet(alpha=2 * reg, l1_ratio=0.5, fit_intercept=bias).fit(X, y)
        assert np.all(np.isclose(b[:3], sk.coef_, atol=1e-4))
        if bias:
            assert np.isclose(float(b[-1]), sk.intercept_, atol=1e-4)


def test_elastic_net_regression(pds):  # :677-715
    from sklearn import linear_model

    X, y = frame3(677)
    for reg in (0.01, 0.05, 0.1, 0.2):
        sk = linear_model.ElasticNet(alpha=2 * reg, l1_ratio=0.5, fit_intercept=False).fit(X, y)
        assert np.all(np.abs(sk.coef_ - pds.lin_reg(*cols(X), target=y, l1_reg=reg, l2_reg=reg, add_bias=False)) < 1e-4)


# ------------------------------------------------------------------------------------------------ recursive / rolling
@pytest.mark.parametrize("l2_reg", [0.0, 0.1])
def test_recursive_lin_reg_and_ridge(pds, l2_reg):  # :718-762, :765-811
    X, y = frame3(718, size=1_000)
    start_with = 3
    co, pred, valid = pds.recursive_lin_reg(*cols(X), target=y, start_with=start_with, l2_reg=l2_reg)
    assert not valid[: start_with - 1].any() and valid[start_with - 1 :].all()
    for i in range(start_with, 30):
        normal = pds.lin_reg(*cols(X[:i]), target=y[:i], l2_reg=l2_reg, singular_x_tol=0.0)
        assert np.all(np.abs(normal - co[i - 1]) < 1e-5)


@pytest.mark.parametrize("l2_reg", [0.0, 0.1])
def test_rolling_lin_reg_and_ridge(pds, l2_reg):  # :814-854, :123-166
    X, y = frame3(814, size=500)
    for w in (5, 8, 12, 15):
        co, pred, valid = pds.rolling_lin_reg(*cols(X), target=y, window_size=w, l2_reg=l2_reg)
        assert not valid[: w - 1].any() and valid[w - 1 :].all()
        for i in range(len(y) - w + 1):
            ans = pds.lin_reg(*cols(X[i : i + w]), target=y[i : i + w], l2_reg=l2_reg, singular_x_tol=0.0)
            np.testing.assert_allclose(co[i + w - 1], ans, rtol=1e-5, atol=1e-8)  # assert_frame_equal defaults


def test_rolling_null_skips(pds):  # :858-908: which rows of the output are null under skip / min_valid_rows
    rng = np.random.default_rng(858)
    size = 1000
    X = rng.random((size, 3))
    nulls = rng.random((size, 3)) < 0.15
    Xn = np.where(nulls, np.nan, X)  # the host layer turns Arrow nulls into NaN for the window kernels
    y = Xn @ [0.15, 0.3, -1.5] + 1e-4 * rng.random(size)
    w, m = 6, 5
    co, pred, valid = pds.rolling_lin_reg(*cols(Xn), target=y, window_size=w, min_valid_rows=m, skip_non_finite=True)
    null_ref = nulls.any(axis=1)
    should_be_null = [True] * (w - 1) + [(w - int(null_ref[i : i + w].sum())) < m for i in range(size - w + 1)]
    assert np.array_equal(~np.asarray(valid, dtype=bool), np.array(should_be_null))


# ------------------------------------------------------------------------------------------------ report
def test_hc_lin_reg_report(pds):  # :169-201 (statsmodels is absent: its HC0-3 formulas restated in NumPy)
    rng = np.random.default_rng(169)
    n = 2_000
    X = rng.random((n, 3))
    y = X @ [0.5, 0.25, -0.15] + rng.normal(size=n) * (0.2 + X[:, 0])
    Xb = np.c_[X, np.ones(n)]
    inv = np.linalg.inv(Xb.T @ Xb)
    beta = inv @ Xb.T @ y
    e = y - Xb @ beta
    h = np.einsum("ij,jk,ik->i", Xb, inv, Xb)
    ref = {"se": np.sqrt(np.diag(inv) * (e @ e) / (n - 4)),
           "hc0": np.sqrt(np.diag(inv @ (Xb.T * e**2) @ Xb @ inv)),
           "hc1": np.sqrt(np.diag(inv @ (Xb.T * e**2) @ Xb @ inv) * n / (n - 4)),
           "hc2": np.sqrt(np.diag(inv @ (Xb.T * (e**2 / (1 - h))) @ Xb @ inv)),
           "hc3": np.sqrt(np.diag(inv @ (Xb.T * (e**2 / (1 - h) ** 2)) @ Xb @ inv))}
    for se, want in ref.items():
        r = pds.lin_reg_report(*cols(X), target=y, add_bias=True, std_err=se)
        got = np.asarray(r["std_err" if se == "se" else f"{se}_se"])
        assert np.all(np.abs(got - want) < 1e-7)


def test_lin_reg_report_already_float64_cast_guard(pds):  # :984-1028
    rng = np.random.default_rng(2)
    n = 300
    x1, x2 = rng.standard_normal(n), rng.standard_normal(n)
    y = 0.5 * x1 - 0.3 * x2 + 0.1 * rng.standard_normal(n)
    rep = pds.lin_reg_report(x1, x2, target=y, add_bias=True)
    rep32 = pds.lin_reg_report(x1.astype(np.float32), x2.astype(np.float32), target=y.astype(np.float32), add_bias=True)
    np.testing.assert_allclose(rep["beta"], rep32["beta"], rtol=1e-6, atol=1e-7)  # f32 inputs are cast up, like the reference
    beta_ref, *_ = np.linalg.lstsq(np.c_[x1, x2, np.ones(n)], y, rcond=None)
    np.testing.assert_allclose(rep["beta"], beta_ref, rtol=1e-10, atol=1e-12)


def test_wls_report_weights(pds):  # :1031-1066 (the chunking of the weights is the plugin layer's business: test_plugin_abi.py)
    rng = np.random.default_rng(3)
    n = 200
    x = rng.standard_normal(n)
    y = 2.0 * x + 0.1 * rng.standard_normal(n)
    w = rng.uniform(0.5, 1.5, n)
    rep = pds.lin_reg_report(x, target=y, weights=w, add_bias=True)
    Xb = np.c_[x, np.ones(n)]
    np.testing.assert_allclose(rep["beta"], np.linalg.solve(Xb.T @ (Xb * w[:, None]), Xb.T @ (w * y)), rtol=1e-10, atol=1e-12)


# ------------------------------------------------------------------------------------------------ singular_x_tol gate suite :1184-1340
def _collinear(n=64, seed=0):  # :1191-1201
    rng = np.random.default_rng(seed)
    x1 = rng.standard_normal(n)
    return x1, 2.0 * x1, rng.standard_normal(n)


def _scaled_singular(n=2000, feats=7, scale=1e3, seed=3):  # :1291-1299
    rng = np.random.default_rng(seed)
    base = rng.standard_normal(n) * scale
    return [base * (i + 1) for i in range(feats)], rng.standard_normal(n) * scale


def _scaled_full_rank(n=2000, feats=7, scale=1e3, seed=4):  # :1302-1308
    rng = np.random.default_rng(seed)
    xs = [rng.standard_normal(n) * scale for _ in range(feats)]
    return xs, sum(xs)


def test_singular_x_tol_nulls_collinear_coeffs(pds, lin_reg_dtype):  # :1204-1208
    x1, x2, y = _collinear()
    assert pds.lin_reg(x1, x2, target=y, add_bias=False) is None


def test_singular_x_tol_off_returns_finite(pds, lin_reg_dtype):  # :1211-1219
    x1, x2, y = _collinear()
    b = pds.lin_reg(x1, x2, target=y, add_bias=False, singular_x_tol=0.0)
    assert b is not None and len(b) == 2


def test_singular_x_tol_well_conditioned_unchanged(pds, lin_reg_dtype):  # :1222-1248
    from sklearn.linear_model import LinearRegression

    rng = np.random.default_rng(7)
    n = 500
    X = rng.standard_normal((n, 3))
    y = X @ [1.0, -0.5, 2.0]
    b = pds.lin_reg(*cols(X), target=y, add_bias=False)
    assert b is not None
    tol = 1e-4 if lin_reg_dtype == "f32" else 1e-9
    np.testing.assert_allclose(b, LinearRegression(fit_intercept=False).fit(X, y).coef_, rtol=tol, atol=tol)


def test_singular_x_tol_group_by_nulls_degenerate_group(pds, lin_reg_dtype):  # :1251-1270
    rng = np.random.default_rng(11)
    n = 40
    good_x1, bad_x1 = rng.standard_normal(n), rng.standard_normal(n)
    x1 = np.concatenate([good_x1, bad_x1])
    x2 = np.concatenate([rng.standard_normal(n), 2.0 * bad_x1])
    y = rng.standard_normal(2 * n)
    keys, co, nu = pds.lin_reg_by_key(x1, x2, target=y, key=np.repeat([0, 1], n))  # 0 = "good", 1 = "bad"
    assert not nu[0] and nu[1] and np.isfinite(co[0]).all()


def test_singular_x_tol_return_pred_nulls(pds, lin_reg_dtype):  # :1273-1280
    x1, x2, y = _collinear(n=32)
    assert pds.lin_reg(x1, x2, target=y, add_bias=False, return_pred=True) is None  # all-null pred / resid


def test_singular_x_tol_multi_target_nulls(pds, lin_reg_dtype):  # :1283-1288
    x1, x2, y = _collinear(n=64)
    s = pds.lin_reg(x1, x2, target=[y, y * 0.5 + 1.0], add_bias=False)
    assert s["target_0"] is None and s["target_1"] is None


def test_singular_x_tol_large_scale_singular_nulls(pds, lin_reg_dtype):  # :1311-1316 (f32: the diagonal product overflows)
    xs, y = _scaled_singular()
    assert pds.lin_reg(*xs, target=y, add_bias=False) is None


def test_singular_x_tol_large_scale_full_rank_not_nulled(pds, lin_reg_dtype):  # :1319-1323
    xs, y = _scaled_full_rank()
    assert pds.lin_reg(*xs, target=y, add_bias=False) is not None


@pytest.mark.parametrize("solver", ["qr", "svd", "choleskey"])
def test_singular_x_tol_per_solver(pds, lin_reg_dtype, solver):  # :1326-1340
    xs, y = _scaled_singular()
    assert pds.lin_reg(*xs, target=y, add_bias=False, solver=solver) is None
    xs2, y2 = _scaled_full_rank()
    assert pds.lin_reg(*xs2, target=y2, add_bias=False, solver=solver) is not None


def test_singular_gate_in_the_grouped_kernel_at_large_scale(pds, lin_reg_dtype):
    """The same overflow regime through the fused grouped kernel (its gate is a product of pivot ratios, taken in f64)."""
    xs, y = _scaled_singular()
    xf, yf = _scaled_full_rank()
    X = [np.concatenate([a, b]) for a, b in zip(xs, xf)]
    keys, co, nu = pds.lin_reg_by_key(*X, target=np.concatenate([y, yf]), key=np.repeat([0, 1], len(y)))
    assert nu[0] and not nu[1]
    np.testing.assert_allclose(co[1], np.ones(7), rtol=1e-3 if lin_reg_dtype == "f32" else 1e-9)


# ------------------------------------------------------------------------------------------------ f32 symbols :204-373
def test_f32_lin_reg_against_sklearn(pds):  # :313-373
    from sklearn import linear_model

    pds.config.LIN_REG_EXPR_F64 = False
    try:
        X, y = frame3(313, beta=(0.5, 0.1, -0.15))
        b = pds.lin_reg(*cols(X), target=y, add_bias=True)
        assert b.dtype == np.float32
        reg = linear_model.LinearRegression(fit_intercept=True).fit(X, y)
        assert np.all(np.abs(b[:3] - reg.coef_) < 1e-4) and abs(b[-1] - reg.intercept_) < 1e-4
        b = pds.lin_reg(*cols(X), target=y, l2_reg=0.1, add_bias=True)
        reg = linear_model.Ridge(alpha=0.1, fit_intercept=True).fit(X, y)
        assert np.all(np.abs(b[:3] - reg.coef_) < 1e-3) and abs(b[-1] - reg.intercept_) < 1e-3
        co, pred, valid = pds.rolling_lin_reg(*cols(X[:500]), target=y[:500], window_size=12)
        assert co.dtype == np.float32 and valid[11:].all()
        r = pds.lin_reg_report(*cols(X), target=y, add_bias=True)
        assert np.asarray(r["beta"]).dtype == np.float32
    finally:
        pds.config.LIN_REG_EXPR_F64 = True


# ------------------------------------------------------------------------------------------------ null policies tests/test_many.py:1636-1726
@pytest.mark.parametrize("policy,fill", [("zero", 0.0), ("one", 1.0), ("0.5", 0.5)])
def test_null_policy_fill_matches_prefilled_frame(pds, policy, fill):
    import pyarrow as pa

    rng = np.random.default_rng(7)
    n = 1000
    X = rng.random((n, 3))
    y = X @ [0.3, -0.2, 0.7] + 0.01 * rng.random(n)
    mask = rng.random(n) < 0.1
    x1 = pa.array(X[:, 0], mask=mask)
    got = pds.lin_reg(x1, pa.array(X[:, 1]), pa.array(X[:, 2]), target=pa.array(y), add_bias=True, null_policy=policy)
    Xf = X.copy()
    Xf[mask, 0] = fill
    np.testing.assert_allclose(got, pds.lin_reg(*cols(Xf), target=y, add_bias=True), atol=1e-8)
    with pytest.raises(Exception, match="Nulls found in data"):
        pds.lin_reg(x1, pa.array(X[:, 1]), pa.array(X[:, 2]), target=pa.array(y), add_bias=True, null_policy="raise")
    with pytest.raises(Exception):
        pds.lin_reg(x1, pa.array(X[:, 1]), pa.array(X[:, 2]), target=pa.array(y), add_bias=True, null_policy="not a policy")
