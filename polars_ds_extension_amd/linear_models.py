"""
Model classes over NumPy / torch data -- the mirror of the reference's pyclass route:
`LR`, `ElasticNet`, `OnlineLR` of /root/reference/python/polars_ds/linear_models.py:134-700, whose `PyLR` /
`PyElasticNet` / `PyOnlineLR` (src/pymodels/py_lr.rs:21-224) call the same solvers as the expressions.  Same
constructor arguments, method names and error behaviour; the fits run on the MI355X through the C ABI
(`lstsq.py`): the row-major feature matrix is transposed once on the device into the column buffers the kernels
stream (`pds_rows_to_cols_*`; a NumPy matrix crosses PCIe as contiguous row chunks, a CUDA tensor stays in HBM --
the reference reads the NumPy buffer through a strided `MatRef`, numpy_faer.rs:10-66).  The O(p'^2) state arithmetic of `OnlineLR.update` (one `woodbury_step`, lr_online_solvers.rs:307-332) and
`predict` (one matrix-vector product) stay where the data is.
`fit_df` / `predict_df` need polars (absent in this image) and are written against its public API.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib, config, lstsq

__all__ = ["LR", "ElasticNet", "OnlineLR", "GLM"]


def _is_torch(a) -> bool:
    return type(a).__module__.startswith("torch")


def _as_matrix(X):
    if _is_torch(X):
        if X.ndim != 2:
            raise ValueError("X must be a 2D matrix.")
        return X
    X = np.asarray(X, dtype=np.float64)  # _sanitize_np (linear_models.py:72-101)
    if X.ndim != 2:
        raise ValueError("X must be a 2D matrix.")
    return X


def _columns(X) -> list:
    """
    The p contiguous columns of an n x p matrix.  With a GPU the row-major matrix is transposed ONCE on the device
    (`pds_rows_to_cols_*`, csrc/layout.hip): a NumPy matrix crosses PCIe as contiguous row chunks -- one copy, not p strided
    host gathers -- and a row-major CUDA tensor never leaves HBM.  The reference reads the NumPy buffer through a strided
    faer MatRef (src/pymodels/numpy_faer.rs:10-66).  Without a visible device the columns are cut on the host (the fit that
    follows then fails in the library: there is no CPU path).
    """
    try:
        import torch

        have_gpu = torch.cuda.is_available()
    except ImportError:
        have_gpu = False
    if have_gpu:
        f64 = bool(config.LIN_REG_EXPR_F64)
        tdt, ndt = (torch.float64, np.float64) if f64 else (torch.float32, np.float32)
        n, p = int(X.shape[0]), int(X.shape[1])
        if n > 0 and p > 0:
            ctx = lstsq.default_context()
            if _is_torch(X):
                Xs = X if X.is_cuda else X.cuda()
                Xs = Xs.to(tdt)
                if Xs.stride(1) != 1:
                    Xs = Xs.contiguous()
                ptr, ld, space, dev = int(Xs.data_ptr()), int(Xs.stride(0)), _lib.PDS_DEVICE, Xs.device
            else:
                Xs = np.ascontiguousarray(X, dtype=ndt)
                ptr, ld, space, dev = int(Xs.ctypes.data), p, _lib.PDS_HOST, torch.device("cuda", ctx.device)
            out = torch.empty((p, n), dtype=tdt, device=dev)
            ctx.follow_torch_stream(dev)
            _lib.check(ctx.fn("pds_rows_to_cols")(ctx._h, C.c_void_p(ptr), C.c_int64(ld), C.c_int64(n), C.c_int(p), C.c_int(space),
                                                  C.c_void_p(int(out.data_ptr())), C.c_int64(n)))
            del Xs
            return [out[j] for j in range(p)]
    if _is_torch(X):
        Xt = X.t().contiguous()
        return [Xt[j] for j in range(Xt.shape[0])]
    return [np.ascontiguousarray(X[:, j]) for j in range(X.shape[1])]


def _fit_rowmajor(X, y, prm, mode: int, pp: int):
    """
    LR / ElasticNet / OnlineLR fits straight from the row-major matrix (`pds_lr_rowmajor_*`): up to 16 features the matrix
    core reads the rows as they lie -- one pass over X, nothing transposed, a NumPy matrix crossing PCIe as contiguous row
    chunks.  Returns (coeffs, is_null, inv) or None when no device is visible, in which case the caller cuts columns.
    """
    try:
        import torch

        if not torch.cuda.is_available():
            return None
    except ImportError:
        return None
    f64 = bool(config.LIN_REG_EXPR_F64)
    tdt, ndt = (torch.float64, np.float64) if f64 else (torch.float32, np.float32)
    n, p = int(X.shape[0]), int(X.shape[1])
    ctx = lstsq.default_context()
    if _is_torch(X):
        Xs = (X if X.is_cuda else X.cuda()).to(tdt)
        if Xs.stride(1) != 1:
            Xs = Xs.contiguous()
        ys = torch.as_tensor(y).to(device=Xs.device, dtype=tdt).reshape(-1).contiguous()
        xp, yp, ld, space = int(Xs.data_ptr()), int(ys.data_ptr()), int(Xs.stride(0)), _lib.PDS_DEVICE
        ctx.follow_torch_stream(Xs.device)
    else:
        Xs = np.ascontiguousarray(X, dtype=ndt)
        ys = np.ascontiguousarray(np.asarray(y.cpu() if _is_torch(y) else y).reshape(-1), dtype=ndt)
        xp, yp, ld, space = int(Xs.ctypes.data), int(ys.ctypes.data), p, _lib.PDS_HOST
    co = np.empty(pp, dtype=ndt)
    inv = np.empty((pp, pp), dtype=ndt) if mode == 2 else None
    is_null = C.c_int(0)
    _lib.check(ctx.fn("pds_lr_rowmajor")(ctx._h, C.c_void_p(xp), C.c_int64(ld), C.c_void_p(yp), C.c_int64(n), C.c_int(p), C.c_int(space),
                                         C.byref(prm), C.c_int(mode), C.c_void_p(co.ctypes.data), C.byref(is_null),
                                         C.c_void_p(inv.ctypes.data) if inv is not None else C.c_void_p(None)))
    return co, bool(is_null.value), inv


def _target(y, n: int):
    if _is_torch(y):
        y = y.reshape(-1)
    else:
        y = np.asarray(y, dtype=np.float64).reshape(-1)
    if int(y.shape[0]) != n:
        raise ValueError("Dimension mismatch.")  # LinalgErrors::DimensionMismatch (src/linear/lr/mod.rs:117-119)
    return y


def _handle_nans_in_np(X: np.ndarray, y: np.ndarray, null_policy: str):
    """linear_models.py:104-131, on NaNs of NumPy inputs."""
    if null_policy == "ignore" or _is_torch(X):
        return X, y
    y2 = y.reshape((-1, 1))
    if null_policy == "raise":
        if np.any(np.isnan(X)) | np.any(np.isnan(y2)):
            raise ValueError("Nulls found in X or y.")
        return X, y
    if null_policy == "skip":
        bad = np.any(np.isnan(X), axis=1) | np.any(np.isnan(y2), axis=1)
        return X[~bad], y[~bad]
    fill = {"zero": 0.0, "one": 1.0}.get(null_policy)
    if fill is None:
        try:
            fill = float(null_policy)
            if not np.isfinite(fill):
                raise ValueError("When null_policy is a number, it cannot be nan or infinite.")
        except Exception as e:
            raise ValueError(f"Unknown null_policy. Error: {e}")
    y_nans = np.any(np.isnan(y2), axis=1)
    return np.nan_to_num(X, nan=fill)[~y_nans], y[~y_nans]


def _predict(X, coeffs: np.ndarray, bias: float):
    X = _as_matrix(X)
    if int(X.shape[1]) != len(coeffs):
        raise ValueError("Dimension mismatch.")
    if _is_torch(X):
        import torch

        b = torch.as_tensor(coeffs, dtype=X.dtype, device=X.device)
        return (X @ b + bias).reshape(-1, 1)
    return (X @ coeffs + bias).reshape(-1, 1)


class _Fitted:
    """coefficients + bias with the accessors the three classes share (LinearModel trait, src/linear/lr/mod.rs:81-112)."""

    def __init__(self, has_bias: bool, feature_names_in_: Sequence[str] | None):
        self._has_bias = bool(has_bias)
        self._coeffs: np.ndarray | None = None
        self._bias = 0.0
        self.feature_names_in_: List[str] = [] if feature_names_in_ is None else list(feature_names_in_)

    def is_fit(self) -> bool:
        return self._coeffs is not None

    def set_input_features(self, features: List[str]):
        self.feature_names_in_ = list(features)
        return self

    def coeffs(self) -> np.ndarray:
        """A copy of the coefficients (without the bias)."""
        return np.array([] if self._coeffs is None else self._coeffs, dtype=np.float64)

    def bias(self) -> float:
        return float(self._bias)

    def _take(self, all_coeffs):
        a = np.asarray(all_coeffs.cpu() if _is_torch(all_coeffs) else all_coeffs, dtype=np.float64).reshape(-1)
        if self._has_bias:
            self._coeffs, self._bias = a[:-1].copy(), float(a[-1])
        else:
            self._coeffs, self._bias = a.copy(), 0.0

    def _set_values(self, coeffs, bias: float):
        self._coeffs = np.ascontiguousarray(coeffs, dtype=np.float64).flatten()
        self._bias = float(bias)
        self._has_bias = bias != 0.0

    def predict(self, X):
        """The prediction of this linear model, an n x 1 matrix."""
        if not self.is_fit():
            raise ValueError("Matrix is not learned yet.")  # LinalgErrors::MatNotLearnedYet
        return _predict(X, self._coeffs, self._bias)

    def predict_df(self, df, name: str = "prediction"):
        if len(self.feature_names_in_) <= 0:
            raise ValueError(
                "The linear model is not fitted on a dataframe, or no feature names have been given."
                "Not enough info to predict on a dataframe. Hint: try .fit_df() or .set_input_features()."
            )
        import polars as pl

        pred = pl.sum_horizontal(beta * pl.col(c) for c, beta in zip(self.feature_names_in_, self._coeffs))
        if self._bias != 0.0:
            pred = pred + self._bias
        return df.with_columns(pred.alias(name))

    def _frame_to_numpy(self, df, features: List[str], target: str, null_policy: str):
        import polars as pl

        lf = df.lazy()
        if null_policy == "skip":
            lf = lf.drop_nulls(subset=list(features) + [target])
        elif null_policy not in ("ignore", "raise"):
            fill = {"zero": 0.0, "one": 1.0}.get(null_policy)
            if fill is None:
                fill = float(null_policy)
                if not np.isfinite(fill):
                    raise ValueError("When null_policy is a number, it cannot be nan or infinite.")
            lf = lf.with_columns(pl.col(features).fill_null(fill)).drop_nulls(subset=target)
        df2 = lf.select(*features, target).collect()
        if null_policy == "raise" and any(df2[c].has_nulls() for c in df2.columns):
            raise ValueError("Nulls found in Dataframe.")
        return df2.select(features).to_numpy(), df2.select(target).to_numpy()


class LR(_Fitted):
    """Normal or Ridge Regression (linear_models.py:134-347; LR::fit_unchecked = faer_solve_lr, lr_solvers.rs:65-73)."""

    def __init__(self, has_bias: bool = False, lambda_: float = 0.0, solver: str = "qr", feature_names_in_: List[str] | None = None):
        super().__init__(has_bias, feature_names_in_)
        self.lambda_ = float(lambda_)
        self.solver = solver

    @classmethod
    def from_values(cls, coeffs: List[float], bias: float = 0.0, feature_names_in_: List[str] | None = None):
        lr = cls(has_bias=(bias != 0.0), lambda_=0.0, solver="Not Solved", feature_names_in_=feature_names_in_)
        lr._set_values(coeffs, bias)
        return lr

    def __repr__(self) -> str:
        out = "Linear Regression (Ridge) Model\n" if self.lambda_ > 0.0 else "Linear Regression Model\n"
        if self.is_fit():
            return out + f"Coefficients: {list(round(float(x), 5) for x in self._coeffs)}\nBias/Intercept: {self._bias}\n"
        return out + "Not fitted yet."

    def fit(self, X, y, null_policy: str = "ignore"):
        X = _as_matrix(X)
        y = _target(y, int(X.shape[0]))
        X, y = _handle_nans_in_np(X, y, null_policy)
        n, p = int(X.shape[0]), int(X.shape[1])
        if n < p or n == 0:
            raise ValueError("Not enough data.")  # LinalgErrors::NotEnoughData (src/linear/lr/mod.rs:119-121)
        # faer_solve_lr: no rank gate, lambda on the feature diagonals only
        prm = lstsq._params(self._has_bias, 0.0, self.lambda_, 1e-5, self.solver, False, 200, 0.0)
        rm = _fit_rowmajor(X, y, prm, 0, p + int(self._has_bias))
        if rm is not None:
            self._take(rm[0])
            return self
        b = lstsq.lin_reg(*_columns(X), target=y, add_bias=self._has_bias, l2_reg=self.lambda_, solver=self.solver,
                          singular_x_tol=0.0, null_policy="ignore")
        self._take(b)
        return self

    def fit_df(self, df, features: List[str], target: str, null_policy: str = "skip", show_report: bool = False):
        X, y = self._frame_to_numpy(df, features, target, null_policy)
        self.feature_names_in_ = list(features)
        if show_report and self.lambda_ == 0.0:
            rep = lstsq.lin_reg_report(*_columns(X), target=y.reshape(-1), add_bias=self._has_bias)
            print({k: np.asarray(v) for k, v in rep.items()})
        return self.fit(X, y, null_policy="ignore")


class ElasticNet(_Fitted):
    """Elastic net by coordinate descent (linear_models.py:350-557; ElasticNet::fit_unchecked = faer_coordinate_descent)."""

    def __init__(self, l1_reg: float, l2_reg: float, has_bias: bool = False, tol: float = 1e-5, max_iter: int = 2000,
                 feature_names_in_: List[str] | None = None):
        if l1_reg <= 0.0 and l2_reg <= 0.0:
            raise ValueError("Cannot have both l1_reg and l2_reg <= 0.")
        super().__init__(has_bias, feature_names_in_)
        self.l1_reg, self.l2_reg, self.tol, self.max_iter = float(l1_reg), float(l2_reg), float(tol), int(max_iter)

    @classmethod
    def from_values(cls, coeffs: List[float], bias: float = 0.0, feature_names_in_: List[str] | None = None):
        en = cls.__new__(cls)
        _Fitted.__init__(en, bias != 0.0, feature_names_in_)
        en.l1_reg = en.l2_reg = float("nan")
        en.tol, en.max_iter = 1e-5, 2000
        en._set_values(coeffs, bias)
        return en

    def has_bias(self) -> bool:
        return self._has_bias

    def __repr__(self) -> str:
        out = "Elastic Net Model\n"
        if self.is_fit():
            return out + f"Coefficients: {list(round(float(x), 5) for x in self._coeffs)}\nBias/Intercept: {self._bias}\n"
        return out + "Not fitted yet."

    def fit(self, X, y, null_policy: str = "ignore"):
        X = _as_matrix(X)
        y = _target(y, int(X.shape[0]))
        X, y = _handle_nans_in_np(X, y, null_policy)
        if int(X.shape[0]) == 0:
            raise ValueError("Not enough data.")  # (fewer rows than columns is fine here, lr_solvers.rs:167-175)
        # always coordinate descent, like ElasticNet::fit_unchecked -- also for a pure ridge penalty (l1_reg <= 0), whose
        # coordinate-descent objective penalises with n_rows * l2_reg (lr_solvers.rs:478-480), not the closed form's l2_reg
        prm = lstsq._params(self._has_bias, self.l1_reg, self.l2_reg, self.tol, "qr", False, self.max_iter, 0.0)
        rm = _fit_rowmajor(X, y, prm, 1, int(X.shape[1]) + int(self._has_bias))
        if rm is not None:
            self._take(rm[0])
            return self
        b = lstsq.elastic_net_fit(*_columns(X), target=y, add_bias=self._has_bias, l1_reg=self.l1_reg, l2_reg=self.l2_reg,
                                  tol=self.tol, max_iter=self.max_iter)
        self._take(b)
        return self

    def fit_df(self, df, features: List[str], target: str, null_policy: str = "skip"):
        X, y = self._frame_to_numpy(df, features, target, null_policy)
        self.feature_names_in_ = list(features)
        return self.fit(X, y, null_policy="ignore")


class OnlineLR(_Fitted):
    """
    Normal or ridge online regression (linear_models.py:560-700).  `fit` is faer_qr_lr_with_inv on the device
    (pds_lr_with_inv_*), `update` one woodbury_step on the p' x p' state; rows holding a NaN are ignored by `update`.
    """

    def __init__(self, lambda_: float = 0.0, has_bias: bool = False):
        super().__init__(has_bias, None)
        self.lambda_ = float(lambda_)
        self._inv: np.ndarray | None = None
        self._all: np.ndarray | None = None  # coefficients incl. the bias (the woodbury state)

    @classmethod
    def from_coeffs_bias_inverse(cls, coeffs: List[float], bias: float, inv: np.ndarray):
        coefficients = np.ascontiguousarray(coeffs, dtype=np.float64).flatten()
        inv = np.asarray(inv, dtype=np.float64)
        if len(coefficients) != inv.shape[1]:
            raise ValueError("Dimension mismatch.")  # set_coeffs_bias_inverse, lr_online_solvers.rs:35-37
        lr = cls(has_bias=(bias > 0.0), lambda_=0.0)
        lr._has_bias = abs(bias) > np.finfo(np.float64).eps
        lr._all = np.r_[coefficients, bias] if lr._has_bias else coefficients.copy()
        lr._inv = inv.copy()
        lr._take(lr._all)
        return lr

    def __repr__(self) -> str:
        out = "Online Linear Regression (Ridge) Model\n" if self.lambda_ > 0.0 else "Online Linear Regression Model\n"
        if self.is_fit():
            return out + f"Coefficients: {list(round(float(x), 5) for x in self._coeffs)}\nBias/Intercept: {self._bias}\n"
        return out + "Not fitted yet."

    def inv(self) -> np.ndarray:
        """A copy of the current inverse of X'X (+ lambda)."""
        if self._inv is None:
            raise ValueError("Matrix is not learned yet.")
        return self._inv.copy()

    def fit(self, X, y):
        X = _as_matrix(X)
        y = _target(y, int(X.shape[0]))
        if not _is_torch(X) and (np.any(np.isnan(X)) | np.any(np.isnan(y))):
            raise ValueError("Online regression currently must fit without null for the initial fit.")
        n, p = int(X.shape[0]), int(X.shape[1])
        if n < p or n == 0:
            raise ValueError("Not enough data.")
        pp = p + int(self._has_bias)
        prm = lstsq._params(self._has_bias, 0.0, self.lambda_, 1e-5, "qr", False, 200, 0.0)
        rm = _fit_rowmajor(X, y, prm, 2, pp)
        if rm is not None:
            self._all = rm[0].astype(np.float64)
            self._inv = rm[2].astype(np.float64).reshape(pp, pp, order="F")
            self._take(self._all)
            return self
        ctx = lstsq.default_context()
        cols = lstsq._Cols(y, _columns(X))
        lstsq._follow(ctx, cols)
        dt = np.float64 if config.LIN_REG_EXPR_F64 else np.float32
        co, inv = np.empty(pp, dtype=dt), np.empty((pp, pp), dtype=dt)
        lam = C.c_double(self.lambda_) if config.LIN_REG_EXPR_F64 else C.c_float(self.lambda_)
        _lib.check(ctx.fn("pds_lr_with_inv")(ctx._h, cols.cols, cols.n_feat, C.c_int64(cols.n_rows), cols.space, int(self._has_bias), lam,
                                             C.c_void_p(co.ctypes.data), C.c_void_p(inv.ctypes.data)))
        self._all = co.astype(np.float64)
        self._inv = inv.astype(np.float64)
        self._take(self._all)
        return self

    def update(self, X, y, c: float = 1.0):
        if not self.is_fit():
            raise ValueError("You cannot update before the initial fit of the matrix.")
        x = np.asarray(X.cpu() if _is_torch(X) else X, dtype=np.float64).reshape(-1)
        yv = float(np.asarray(y.cpu() if _is_torch(y) else y, dtype=np.float64).reshape(-1)[0])
        if not (np.all(np.isfinite(x)) and np.isfinite(yv)):
            return self  # OnlineLR::update, lr_online_solvers.rs:85-89
        if self._has_bias:
            x = np.r_[x, 1.0]
        if len(x) != len(self._all):
            raise ValueError("Dimension mismatch.")
        # woodbury_step (lr_online_solvers.rs:307-332)
        u = self._inv @ x
        z = 1.0 / (c + float(x @ u))
        self._inv -= z * np.outer(u, u)
        self._all = self._all + u * (z * (yv - float(x @ self._all)))
        self._take(self._all)
        return self


GLM_FAMILIES = {"gaussian": (0, 0), "normal": (0, 0), "poisson": (1, 1), "binomial": (2, 2), "logistic": (2, 2), "gamma": (3, 3)}
"""family -> (link, variance) ids of include/pds_lstsq.h: canonical links, GLMFamily::link_function / variance_function
(src/linear/glm/glm_solvers.rs:24-41)."""


class GLM:
    """
    Generalized linear models by iteratively re-weighted least squares (linear_models.py:705-923 of the reference; PyGLM,
    src/pymodels/py_glm.rs:14-101; faer_irls, src/linear/glm/glm_solvers.rs:249-368 -- the caller of faer_weighted_lr).
    Families and their canonical links: gaussian / normal (identity), poisson (log), binomial / logistic (logit), gamma
    (inverse).  On the MI355X one IRLS iteration is ONE pass over the frame (`pds_glm_irls_*`): weights and working response
    are formed from the previous coefficients while a row sits in registers (up to 16 features; wider frames write them as two
    columns in front of the weighted wide Gram build).
    """

    def __init__(self, add_bias: bool = False, solver: str = "irls", family: str = "normal", max_iter: int = 100, tol: float = 1e-8,
                 feature_names_in_: List[str] | None = None):
        if solver not in ["irls"]:
            raise NotImplementedError
        if max_iter < 1:
            raise ValueError("`max_iter` must be > 1.")
        if family not in ["gaussian", "normal", "poisson", "binomial", "logistic", "gamma"]:
            raise NotImplementedError
        self.add_bias, self.family, self.solver = bool(add_bias), family, solver
        self.max_iter, self.tol = int(max_iter), abs(float(tol))
        self.feature_names_in_: List[str] = [] if feature_names_in_ is None else list(feature_names_in_)
        self._coeffs: np.ndarray | None = None
        self._bias = 0.0
        self.n_iter_ = 0

    def __repr__(self) -> str:
        link = {0: "Identity", 1: "Log", 2: "Logit", 3: "Inverse"}[GLM_FAMILIES[self.family][0]]
        var = {0: "Gaussian", 1: "Poisson", 2: "Binomial", 3: "Gamma"}[GLM_FAMILIES[self.family][1]]
        return f"GLM:\nLink: {link}\nVariance: {var}"  # GLM::to_string, glm_solvers.rs:98-102

    def is_fit(self) -> bool:
        return self._coeffs is not None

    def set_input_features(self, features: List[str]):
        self.feature_names_in_ = list(features)
        return self

    def coeffs(self) -> np.ndarray:
        if self._coeffs is None:
            raise ValueError("Matrix is not learned yet.")
        return self._coeffs.copy()

    def bias(self) -> float:
        return float(self._bias)

    def fit(self, X, y, null_policy: str = "ignore"):
        X = _as_matrix(X)
        y = _target(y, int(X.shape[0]))
        X, y = _handle_nans_in_np(X, y, null_policy)
        n, p = int(X.shape[0]), int(X.shape[1])
        if n < p or n == 0:
            raise ValueError("Not enough data.")  # LinearModel::fit, src/linear/lr/mod.rs:114-125
        ctx = lstsq.default_context()
        cols = lstsq._Cols(y, _columns(X))
        lstsq._follow(ctx, cols)
        pp = p + int(self.add_bias)
        f64 = bool(config.LIN_REG_EXPR_F64)
        co = np.empty(pp, dtype=np.float64 if f64 else np.float32)
        link, var = GLM_FAMILIES[self.family]
        n_iter = C.c_int(0)
        tol = C.c_double(self.tol) if f64 else C.c_float(self.tol)
        _lib.check(ctx.fn("pds_glm_irls")(ctx._h, cols.cols, cols.n_feat, C.c_int64(cols.n_rows), cols.space, int(self.add_bias),
                                          C.c_int(link), C.c_int(var), tol, C.c_int(self.max_iter), C.c_void_p(co.ctypes.data),
                                          C.byref(n_iter)))
        co = co.astype(np.float64)
        self._coeffs, self._bias = (co[:-1].copy(), float(co[-1])) if self.add_bias else (co.copy(), 0.0)
        self.n_iter_ = int(n_iter.value)
        return self

    def fit_df(self, df, features: List[str], target: str, null_policy: str = "skip", show_report: bool = False):
        X, y = _Fitted._frame_to_numpy(self, df, features, target, null_policy)
        self.feature_names_in_ = list(features)
        return self.fit(X, y, null_policy="ignore")

    def predict(self, X, linear: bool = False):
        """E[Y | X] = g^-1(X beta + bias), or the linear predictor eta when `linear` (glm_predict, glm_solvers.rs:243-250)."""
        if not self.is_fit():
            raise ValueError("Matrix is not learned yet.")
        eta = _predict(X, self._coeffs, self._bias)
        if linear:
            return eta
        link = GLM_FAMILIES[self.family][0]
        if _is_torch(eta):
            import torch

            return {0: lambda e: e, 1: torch.exp, 2: torch.sigmoid, 3: torch.reciprocal}[link](eta)
        if link == 2:
            e = np.exp(eta)
            return e / (1.0 + e)
        return {0: lambda e: e, 1: np.exp, 3: lambda e: 1.0 / e}[link](eta)
