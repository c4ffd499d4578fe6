"""
The model classes (polars_ds_extension_amd/linear_models.py), mirroring /root/reference/tests/test_linear_models.py:
`_handle_nans_in_np` (:9-50), LR against scikit-learn for every solver string (:53-75), OnlineLR fit + updates against
refits (:78-122), ElasticNet against scikit-learn with and without intercept (:125-158), GLM for its four families
(:199-257; scikit-learn's unpenalised GLMs in the place of statsmodels, which this image does not have) -- same tolerances.
"""
import numpy as np
import pytest


def _frame(seed, size=5000, noise=1e-4):
    rng = np.random.default_rng(seed)
    X = rng.random((size, 3))
    y = X[:, 0] + 0.2 * X[:, 1] - 0.3 * X[:, 2] + noise * rng.random(size)
    return X, y.reshape(-1, 1)


def test_lr_null_policies_for_np():
    from polars_ds_extension_amd.linear_models import _handle_nans_in_np

    X, y = _frame(0)
    nulls = X[:, 0] > 0.5
    x = X.copy()
    x[nulls, 0] = np.nan
    x_nan, _ = _handle_nans_in_np(x, y, "ignore")
    assert np.all(np.isnan(x_nan[nulls][:, 0]))
    with pytest.raises(ValueError, match="Nulls found in X or y."):
        _handle_nans_in_np(x, y, "raise")
    x_skipped, y_skipped = _handle_nans_in_np(x, y, "skip")
    assert np.all(x_skipped == x[~nulls]) and len(y_skipped) == (~nulls).sum()
    x_zeroed, _ = _handle_nans_in_np(x, y, "zero")
    assert np.all(x_zeroed[nulls][:, 0] == 0.0)
    x_one, _ = _handle_nans_in_np(x, y, "one")
    assert np.all(x_one[nulls][:, 0] == 1.0)
    x_q, _ = _handle_nans_in_np(x, y, "0.25")
    assert np.all(x_q[nulls][:, 0] == 0.25)
    with pytest.raises(ValueError, match="Unknown null_policy"):
        _handle_nans_in_np(x, y, "whatever")


def test_constructors_and_state_without_gpu():
    from polars_ds_extension_amd.linear_models import LR, ElasticNet, OnlineLR

    lr = LR.from_values([1.0, 2.0], bias=0.5, feature_names_in_=["a", "b"])
    assert lr.is_fit() and lr.bias() == 0.5 and list(lr.coeffs()) == [1.0, 2.0]
    np.testing.assert_allclose(lr.predict(np.array([[1.0, 1.0], [0.0, 2.0]])).flatten(), [3.5, 4.5])
    assert "Linear Regression Model" in repr(lr) and not LR().is_fit() and "Not fitted yet." in repr(LR())
    with pytest.raises(ValueError, match="Cannot have both l1_reg and l2_reg <= 0."):
        ElasticNet(0.0, 0.0)
    en = ElasticNet.from_values([0.5], bias=0.0)
    assert en.is_fit() and not en.has_bias()
    o = OnlineLR.from_coeffs_bias_inverse([1.0, -1.0], 0.0, np.eye(2))
    o.update(np.array([1.0, 0.0]), 3.0)  # woodbury_step on the identity: inv -> I - e1 e1' / 2, w -> w + e1 * (3 - 1) / 2
    np.testing.assert_allclose(o.inv(), [[0.5, 0.0], [0.0, 1.0]])
    np.testing.assert_allclose(o.coeffs(), [2.0, -1.0])
    o.update(np.array([np.nan, 0.0]), 3.0)  # rows holding a NaN are ignored
    np.testing.assert_allclose(o.coeffs(), [2.0, -1.0])
    with pytest.raises(ValueError, match="You cannot update before the initial fit"):
        OnlineLR().update(np.array([1.0]), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["svd", "cholesky", "qr"])
def test_lr(solver):
    from sklearn.linear_model import LinearRegression, Ridge

    from polars_ds_extension_amd.linear_models import LR

    X, y = _frame(1)
    ols = LR(False, 0.0, solver).fit(X, y)
    sk = LinearRegression(fit_intercept=False).fit(X, y)
    assert np.all(np.abs(ols.coeffs() - sk.coef_) < 1e-6)
    np.testing.assert_allclose(ols.predict(X[:7]).flatten(), sk.predict(X[:7]).flatten(), atol=1e-6)
    rb = LR(True, 0.7, solver).fit(X, y)  # ridge with intercept: lambda on the features only
    skr = Ridge(alpha=0.7, fit_intercept=True).fit(X, y.ravel())
    assert np.all(np.abs(rb.coeffs() - skr.coef_) < 1e-6) and abs(rb.bias() - skr.intercept_) < 1e-6
    Xn = X.copy()
    Xn[::9, 1] = np.nan
    sk2 = LinearRegression(fit_intercept=False).fit(X[np.isfinite(Xn).all(axis=1)], y[np.isfinite(Xn).all(axis=1)])
    assert np.all(np.abs(LR().fit(Xn, y, null_policy="skip").coeffs() - sk2.coef_) < 1e-6)
    with pytest.raises(ValueError, match="Not enough data."):
        LR().fit(X[:2], y[:2])


@pytest.mark.gpu
def test_lr_on_device_tensors():
    import torch

    from polars_ds_extension_amd.linear_models import LR

    X, y = _frame(2)
    b_host = LR(True).fit(X, y)
    b_dev = LR(True).fit(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda())
    np.testing.assert_allclose(b_dev.coeffs(), b_host.coeffs(), rtol=1e-10)
    assert abs(b_dev.bias() - b_host.bias()) < 1e-10
    pred = b_dev.predict(torch.from_numpy(X[:5]).cuda())
    np.testing.assert_allclose(pred.cpu().numpy(), b_host.predict(X[:5]), rtol=1e-10)


@pytest.mark.gpu
def test_online_lr():
    from sklearn.linear_model import LinearRegression

    from polars_ds_extension_amd.linear_models import OnlineLR

    X, y = _frame(3)
    olr = OnlineLR().fit(X[:10], y[:10])
    sk = LinearRegression(fit_intercept=False).fit(X[:10], y[:10])
    assert np.all(np.abs(olr.predict(X[:10]).flatten() - sk.predict(X[:10]).flatten()) < 1e-6)
    assert np.all(np.abs(olr.coeffs() - sk.coef_) < 1e-6)
    np.testing.assert_allclose(olr.inv(), np.linalg.inv(X[:10].T @ X[:10]), rtol=1e-8)
    for i in range(10, 20):
        olr.update(X[i], y[i])
        sk = LinearRegression(fit_intercept=False).fit(X[: i + 1], y[: i + 1])
        assert np.all(np.abs(olr.coeffs() - sk.coef_) < 1e-6)
    ob = OnlineLR(lambda_=0.3, has_bias=True).fit(X[:50], y[:50])
    Xb = np.c_[X[:50], np.ones(50)]
    G = Xb.T @ Xb + np.diag([0.3, 0.3, 0.3, 0.0])
    np.testing.assert_allclose(np.r_[ob.coeffs(), ob.bias()], np.linalg.solve(G, Xb.T @ y[:50].ravel()), rtol=1e-8)
    ob.update(X[50], y[50], c=1.0).update(X[50], y[50], c=-1.0)  # add a row, then remove it again
    np.testing.assert_allclose(np.r_[ob.coeffs(), ob.bias()], np.linalg.solve(G, Xb.T @ y[:50].ravel()), rtol=1e-7)
    with pytest.raises(ValueError, match="must fit without null"):
        OnlineLR().fit(np.where(X[:10] > 0.9, np.nan, X[:10]), y[:10])


@pytest.mark.gpu
@pytest.mark.parametrize("add_bias", [False, True])
def test_elastic_net(add_bias):
    import sklearn.linear_model as lm

    from polars_ds_extension_amd.linear_models import ElasticNet

    l1_reg = l2_reg = 0.1
    X, y = _frame(4, noise=0.0)
    en = ElasticNet(l1_reg=l1_reg, l2_reg=l2_reg, has_bias=add_bias).fit(X, y)
    sk = lm.ElasticNet(alpha=l1_reg + l2_reg, l1_ratio=l1_reg / (l1_reg + l2_reg), fit_intercept=add_bias).fit(X, y.ravel())
    assert np.all(np.abs(en.coeffs() - sk.coef_) < 1e-4)
    if add_bias:
        assert abs(en.bias() - sk.intercept_) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("add_bias", [False, True])
def test_elastic_net_pure_ridge_penalty_runs_coordinate_descent(add_bias, orc):
    """
    l1_reg <= 0 < l2_reg is a legal ElasticNet (the constructor only rejects both <= 0).  ElasticNet::fit_unchecked
    (lr_solvers.rs:139-164) still runs faer_coordinate_descent, whose norms are X'X_jj + n * l2_reg (:478-480): the answer is
    ridge with n * l2_reg, NOT the closed form of `lin_reg(l2_reg=...)`.
    """
    from polars_ds_extension_amd.linear_models import ElasticNet

    rng = np.random.default_rng(12)
    n = 200
    X = rng.normal(size=(n, 3))
    y = X @ [1.0, 2.0, -1.0] + (0.4 if add_bias else 0.0) + 0.05 * rng.normal(size=n)
    en = ElasticNet(l1_reg=0.0, l2_reg=0.1, has_bias=add_bias, tol=1e-9, max_iter=5000).fit(X, y)
    Xb = np.c_[X, np.ones(n)] if add_bias else X
    ref = orc.coordinate_descent(Xb, y, 0.0, 0.1, add_bias, 1e-9, 5000, False)
    got = np.r_[en.coeffs(), en.bias()] if add_bias else en.coeffs()
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-10
    # ... which is the closed-form ridge with lambda = n * l2_reg, and visibly not the one with lambda = l2_reg
    G = Xb.T @ Xb + np.diag([n * 0.1] * 3 + ([0.0] if add_bias else []))
    assert np.linalg.norm(got - np.linalg.solve(G, Xb.T @ y)) / np.linalg.norm(ref) < 1e-7
    G1 = Xb.T @ Xb + np.diag([0.1] * 3 + ([0.0] if add_bias else []))
    assert np.linalg.norm(got - np.linalg.solve(G1, Xb.T @ y)) / np.linalg.norm(ref) > 1e-2


# ------------------------------------------------------------------------------------------ fits from a row-major matrix
@pytest.mark.gpu
@pytest.mark.parametrize("p", [1, 3, 8, 16, 20])
def test_fits_from_row_major_matrices(p, orc):
    """pds_lr_rowmajor_* (the pyclass route): p <= 16 reads the rows as they lie (moments_rowmajor_kernel), wider matrices are
    transposed on the device; NumPy (host, contiguous row chunks) and CUDA tensors with a row stride > p; ragged row counts."""
    import torch

    import polars_ds_extension_amd as pds
    from polars_ds_extension_amd import _lib
    from polars_ds_extension_amd.linear_models import LR, ElasticNet, OnlineLR

    lib = _lib.load()
    rng = np.random.default_rng(50 + p)
    try:
        for n, chunk_mb in ((2 * p + 3, 256.0), (4099, 256.0), (60_001, 0.05)):
            lib.pds_set_host_staging(chunk_mb, 0.0)
            X = rng.normal(size=(n, p))
            y = X @ rng.normal(size=p) + 0.4 + 0.1 * rng.normal(size=n)
            wide = torch.zeros((n, p + 5), dtype=torch.float64, device="cuda")
            wide[:, :p] = torch.from_numpy(X).cuda()
            Xd = wide[:, :p]  # row stride p + 5
            assert Xd.stride(0) == p + 5 and Xd.stride(1) == 1
            for bias in (False, True):
                bo = orc.pl_lr(X, y, add_bias=bias, singular_x_tol=0.0)
                for data in (X, Xd):
                    lr = LR(has_bias=bias).fit(data, y)
                    b = np.r_[lr.coeffs(), lr.bias()] if bias else lr.coeffs()
                    assert np.linalg.norm(b - bo) / np.linalg.norm(bo) < 1e-9, (n, p, bias)
                    lr = LR(has_bias=bias, lambda_=0.3).fit(data, y)
                    b = np.r_[lr.coeffs(), lr.bias()] if bias else lr.coeffs()
                    br = orc.pl_lr(X, y, add_bias=bias, l2_reg=0.3, singular_x_tol=0.0)
                    assert np.linalg.norm(b - br) / np.linalg.norm(br) < 1e-9
                    en = ElasticNet(l1_reg=0.01, l2_reg=0.02, has_bias=bias, tol=1e-9, max_iter=5000).fit(data, y)
                    b = np.r_[en.coeffs(), en.bias()] if bias else en.coeffs()
                    be = orc.pl_lr(X, y, add_bias=bias, l1_reg=0.01, l2_reg=0.02, tol=1e-9, max_iter=5000)
                    assert np.linalg.norm(b - be) / np.linalg.norm(be) < 1e-7, (n, p, bias, b, be)
                    ol = OnlineLR(has_bias=bias, lambda_=0.1).fit(data, y)
                    Xb = np.c_[X, np.ones(n)] if bias else X
                    G = Xb.T @ Xb
                    G[np.arange(p), np.arange(p)] += 0.1  # lambda on the feature diagonals
                    assert np.linalg.norm(ol.inv() - np.linalg.inv(G)) / np.linalg.norm(np.linalg.inv(G)) < 1e-8
                    bi = np.linalg.solve(G, Xb.T @ y)
                    b = np.r_[ol.coeffs(), ol.bias()] if bias else ol.coeffs()
                    assert np.linalg.norm(b - bi) / np.linalg.norm(bi) < 1e-8
        # f32 twin
        pds.config.LIN_REG_EXPR_F64 = False
        X = rng.normal(size=(30_001, p))
        y = X @ rng.normal(size=p) + 0.4 + 0.1 * rng.normal(size=30_001)
        lr = LR(has_bias=True).fit(X, y)
        bo = orc.pl_lr(X.astype(np.float32).astype(np.float64), y.astype(np.float32).astype(np.float64), add_bias=True, singular_x_tol=0.0)
        assert np.linalg.norm(np.r_[lr.coeffs(), lr.bias()] - bo) / np.linalg.norm(bo) < 1e-4
    finally:
        pds.config.LIN_REG_EXPR_F64 = True
        lib.pds_set_host_staging(256.0, 98304.0)


@pytest.mark.gpu
def test_new_entry_points_reject_bad_arguments():
    """pds_lr_rowmajor_* / pds_glm_irls_* / pds_rows_to_cols_*: argument errors come back as codes + messages, nothing runs."""
    import ctypes as C

    import torch

    from polars_ds_extension_amd import _lib, lstsq

    lib = _lib.load()
    ctx = lstsq.default_context()
    X = np.ascontiguousarray(np.random.default_rng(0).normal(size=(50, 3)))
    y = np.ascontiguousarray(X @ [1.0, 2.0, 3.0])
    co = np.zeros(4)
    prm = lstsq._params(True, 0.0, 0.0, 1e-5, "qr", False, 200, 0.0)
    null = C.c_int(0)

    def rowmajor(ld=3, n=50, p=3, mode=0, inv=None):
        return lib.pds_lr_rowmajor_f64(ctx._h, C.c_void_p(X.ctypes.data), C.c_int64(ld), C.c_void_p(y.ctypes.data), C.c_int64(n), C.c_int(p),
                                       C.c_int(_lib.PDS_HOST), C.byref(prm), C.c_int(mode), C.c_void_p(co.ctypes.data), C.byref(null),
                                       C.c_void_p(inv))

    assert rowmajor() == 0 and np.allclose(co[:3], [1.0, 2.0, 3.0], atol=1e-9)
    assert rowmajor(ld=2) != 0 and b"stride" in lib.pds_last_error()
    assert rowmajor(mode=2) != 0 and b"inv" in lib.pds_last_error()  # OnlineLR mode needs the inverse buffer
    assert rowmajor(mode=7) != 0
    assert rowmajor(n=0) != 0 and b"Empty" in lib.pds_last_error()
    assert rowmajor(n=2) != 0 and b"#Data < #features" in lib.pds_last_error()
    keep = [np.ascontiguousarray(X[:, j]) for j in range(3)]
    cols = (C.c_void_p * 4)(y.ctypes.data, *[k.ctypes.data for k in keep])
    it = C.c_int(0)

    def glm(link=0, var=0, max_iter=10, p=3, n=50):
        return lib.pds_glm_irls_f64(ctx._h, cols, C.c_int(p), C.c_int64(n), C.c_int(_lib.PDS_HOST), 1, C.c_int(link), C.c_int(var),
                                    C.c_double(1e-8), C.c_int(max_iter), C.c_void_p(co.ctypes.data), C.byref(it))

    assert glm() == 0 and it.value >= 1 and np.allclose(co[:3], [1.0, 2.0, 3.0], atol=1e-7)
    assert glm(link=4) != 0 and glm(var=-1) != 0
    assert glm(max_iter=0) != 0 and b"max_iter" in lib.pds_last_error()
    assert glm(n=0) != 0 and b"Empty" in lib.pds_last_error()
    out = torch.empty((3, 50), dtype=torch.float64, device="cuda")

    def r2c(ld=3, stride=50, n=50):
        return lib.pds_rows_to_cols_f64(ctx._h, C.c_void_p(X.ctypes.data), C.c_int64(ld), C.c_int64(n), C.c_int(3), C.c_int(_lib.PDS_HOST),
                                        C.c_void_p(out.data_ptr()), C.c_int64(stride))

    assert r2c() == 0 and np.array_equal(out.cpu().numpy(), X.T)
    assert r2c(ld=2) != 0 and r2c(stride=49) != 0 and r2c(n=0) != 0


# ------------------------------------------------------------------------------------------ GLM (IRLS)
def _glm_family_data(family, rng, n=500, p=4):
    """tests/test_linear_models.py:199-232 of the reference: the four y generators."""
    X = rng.randn(n, p)
    if family == "gaussian":
        y = X @ np.array([1.0, -0.5, 0.3, 0.8]) + rng.randn(n) * 0.1
    elif family == "binomial":
        y = rng.binomial(1, 1.0 / (1.0 + np.exp(-(X @ np.array([1.0, -0.5, 0.3, 0.8]))))).astype(float)
    elif family == "poisson":
        y = rng.poisson(np.exp(np.clip(X @ np.array([0.5, -0.25, 0.15, 0.4]), -2.0, 2.0))).astype(float)
    else:
        y = rng.gamma(shape=2.0, scale=np.exp(np.clip(X @ np.array([0.3, -0.15, 0.09, 0.24]), -2.0, 2.0)) / 2.0)
    return X, y


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["gaussian", "binomial", "poisson", "gamma"])
def test_glm_family(family):
    """All four families against an independent maximum-likelihood fit (the reference compares with statsmodels at atol 1e-2)."""
    from scipy.optimize import minimize

    from polars_ds_extension_amd.linear_models import GLM

    rng = np.random.RandomState(42)
    X, y = _glm_family_data(family, rng)
    glm = GLM(solver="irls", add_bias=True, family=family, max_iter=200, tol=1e-8)
    glm.fit(X, y.reshape(-1, 1))
    b = np.r_[glm.coeffs(), glm.bias()]
    Xb = np.c_[X, np.ones(len(y))]
    # negative log-likelihood with the canonical link (up to constants): the IRLS fixed point is its stationary point
    nll = {
        "gaussian": lambda t: 0.5 * np.sum((y - Xb @ t) ** 2),
        "binomial": lambda t: np.sum(np.logaddexp(0.0, Xb @ t) - y * (Xb @ t)),
        "poisson": lambda t: np.sum(np.exp(Xb @ t) - y * (Xb @ t)),
        "gamma": lambda t: np.sum(y * (Xb @ t) - np.log(np.maximum(Xb @ t, 1e-300))),  # inverse link: mu = 1 / eta
    }[family]
    res = minimize(nll, b + 0.01, method="Nelder-Mead", options={"xatol": 1e-10, "fatol": 1e-14, "maxiter": 20000, "maxfev": 20000})
    np.testing.assert_allclose(b, res.x, atol=1e-2, err_msg=f"GLM family={family!r}")
    # the gradient of the likelihood vanishes at the fit: X' (y - mu) = 0 for a canonical link (gamma: with the sign of -1/mu)
    mu = glm.predict(X).reshape(-1)
    score = Xb.T @ (y - mu)
    assert np.max(np.abs(score)) < 1e-5 * len(y), score
    assert glm.predict(X, linear=True).shape == (len(y), 1) and 1 <= glm.n_iter_ <= 200
    if family == "binomial":  # and scikit-learn's unpenalised logistic regression agrees
        from sklearn.linear_model import LogisticRegression

        m = LogisticRegression(penalty=None, tol=1e-12, max_iter=1000).fit(X, y)
        np.testing.assert_allclose(b, np.r_[m.coef_.reshape(-1), m.intercept_], atol=1e-4)
    if family == "poisson":
        from sklearn.linear_model import PoissonRegressor

        m = PoissonRegressor(alpha=0, tol=1e-12, max_iter=1000).fit(X, y)
        np.testing.assert_allclose(b, np.r_[m.coef_.reshape(-1), m.intercept_], atol=1e-5)


@pytest.mark.gpu
def test_glm_convergence_failure():
    """GLM with max_iter=1 and tight tol does not crash; returns finite coeffs (:260-276)."""
    from polars_ds_extension_amd.linear_models import GLM

    rng = np.random.RandomState(0)
    n, p = 200, 5
    X = rng.randn(n, p)
    y = (np.exp(np.clip(X @ np.ones(p) * 0.2, -2, 2)) + rng.exponential(0.5, n)).reshape(-1, 1)
    glm = GLM(solver="irls", add_bias=False, family="poisson", max_iter=1, tol=1e-12)
    glm.fit(X, y)
    assert glm.coeffs() is not None and np.all(np.isfinite(glm.coeffs())) and glm.n_iter_ == 1
    with pytest.raises(NotImplementedError):
        GLM(solver="lbfgs")
    with pytest.raises(NotImplementedError):
        GLM(family="tweedie")
    with pytest.raises(ValueError, match="max_iter"):
        GLM(max_iter=0)


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["gaussian", "binomial", "poisson", "gamma"])
@pytest.mark.parametrize("add_bias", [False, True])
def test_glm_matches_the_oracle(family, add_bias, orc):
    """pds_glm_irls_* (one Gram pass per IRLS iteration) against the restated faer_irls, f64 and f32, host and device data."""
    import torch

    import polars_ds_extension_amd as pds
    from polars_ds_extension_amd.linear_models import GLM

    rng = np.random.RandomState(7)
    X, y = _glm_family_data(family, rng, n=20_001, p=4)
    if family == "gamma" and not add_bias:
        y = rng.gamma(shape=2.0, scale=1.0 / np.maximum(0.5 + X @ np.array([0.1, -0.05, 0.03, 0.08]), 0.1) / 2.0)
    bo, it_o = orc.glm_irls(X, y, family, add_bias=add_bias, tol=1e-10, max_iter=100)
    for data in ((X, y), (torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda())):
        glm = GLM(add_bias=add_bias, family=family, max_iter=100, tol=1e-10).fit(*data)
        b = np.r_[glm.coeffs(), glm.bias()] if add_bias else glm.coeffs()
        assert np.linalg.norm(b - bo) / np.linalg.norm(bo) < 1e-9, (family, add_bias, b, bo)
        assert abs(glm.n_iter_ - it_o) <= 1
    pds.config.LIN_REG_EXPR_F64 = False
    try:
        glm = GLM(add_bias=add_bias, family=family, max_iter=100, tol=1e-6).fit(X, y)
        b32 = np.r_[glm.coeffs(), glm.bias()] if add_bias else glm.coeffs()
        o32, _ = orc.glm_irls(X.astype(np.float32), y.astype(np.float32), family, add_bias=add_bias, tol=1e-6, max_iter=100)
        d_gpu = np.linalg.norm(b32 - bo) / np.linalg.norm(bo)
        d_orc = np.linalg.norm(o32 - bo) / np.linalg.norm(bo)
        print(f"glm f32 {family} bias={add_bias}: gpu-truth {d_gpu:.2e}  orc32-truth {d_orc:.2e}")
        assert d_gpu < 1e-4 or d_gpu <= 1.25 * d_orc
    finally:
        pds.config.LIN_REG_EXPR_F64 = True


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["gaussian", "binomial", "poisson", "gamma"])
@pytest.mark.parametrize("p", [17, 40])
def test_glm_beyond_16_features_matches_the_oracle(family, p, orc):
    """More than 16 features: weights / working response as two columns + the weighted wide Gram build, same iteration."""
    import torch

    import polars_ds_extension_amd as pds
    from polars_ds_extension_amd.linear_models import GLM

    rng = np.random.RandomState(11 + p)
    n = 30_011
    X = rng.randn(n, p)
    beta = rng.randn(p) * (0.6 / np.sqrt(p))
    eta = X @ beta + 0.2
    if family == "gaussian":
        y = eta + rng.randn(n) * 0.1
    elif family == "binomial":
        y = rng.binomial(1, 1.0 / (1.0 + np.exp(-eta))).astype(float)
    elif family == "poisson":
        y = rng.poisson(np.exp(np.clip(eta, -2.0, 2.0))).astype(float)
    else:  # inverse link: keep 1 / mu = eta well away from zero
        eta = 1.5 + 0.3 * np.tanh(eta)
        y = rng.gamma(shape=2.0, scale=(1.0 / eta) / 2.0)
    bo, it_o = orc.glm_irls(X, y, family, add_bias=True, tol=1e-10, max_iter=100)
    for data in ((X, y), (torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda())):
        glm = GLM(add_bias=True, family=family, max_iter=100, tol=1e-10).fit(*data)
        b = np.r_[glm.coeffs(), glm.bias()]
        assert np.linalg.norm(b - bo) / np.linalg.norm(bo) < 1e-9, (family, p, np.linalg.norm(b - bo) / np.linalg.norm(bo))
        assert abs(glm.n_iter_ - it_o) <= 1
    pds.config.LIN_REG_EXPR_F64 = False
    try:
        glm = GLM(add_bias=True, family=family, max_iter=100, tol=1e-6).fit(X, y)
        b32 = np.r_[glm.coeffs(), glm.bias()]
        o32, _ = orc.glm_irls(X.astype(np.float32), y.astype(np.float32), family, add_bias=True, tol=1e-6, max_iter=100)
        d_gpu = np.linalg.norm(b32 - bo) / np.linalg.norm(bo)
        d_orc = np.linalg.norm(o32 - bo) / np.linalg.norm(bo)
        print(f"glm f32 {family} p={p}: gpu-truth {d_gpu:.2e}  orc32-truth {d_orc:.2e}")
        assert d_gpu < 1e-4 or d_gpu <= 1.25 * d_orc
    finally:
        pds.config.LIN_REG_EXPR_F64 = True
