"""
Host-side mirror of the reference's least-squares expression interface on top of the C ABI.

Function names, keyword names, defaults and error behaviour follow
/root/reference/python/polars_ds/exprs/expr_linear.py (`lin_reg` :105-274, `lin_reg_report` :561-631,
`rolling_lin_reg` :482-558, `recursive_lin_reg` :413-479) and the Rust plugin functions they call
(src/num_ext/linear_regression.rs).  What differs is only the carrier: instead of a lazy `pl.Expr`
evaluated by Polars (not installable in this image) the functions take the column buffers directly --
numpy arrays (host memory, what an Arrow Float64/Float32 buffer is) or torch CUDA tensors (already in HBM).
All arithmetic happens in libpds_lstsq_hip.so; nothing here computes a regression on the CPU.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Sequence

import numpy as np

from . import _lib, config

__all__ = [
    "Context", "default_context", "lin_reg", "lin_reg_report", "lin_reg_by", "rolling_lin_reg",
    "recursive_lin_reg", "lin_reg_w_rcond", "elastic_net_fit", "report_fit_from_moments", "report_partials", "report_finish", "gram_moments", "lin_reg_from_moments", "query_ar_coeffs",
]


def _is_torch(a) -> bool:
    return type(a).__module__.startswith("torch")


def _dtype():
    return np.float64 if config.LIN_REG_EXPR_F64 else np.float32


def _suffix() -> str:
    return "_f64" if config.LIN_REG_EXPR_F64 else "_f32"


class _Cols:
    """Column buffers (kept alive) + the void** array the C ABI wants, in the order [y, x1..xp]."""

    def __init__(self, y, xs: Sequence, weights=None):
        dt = _dtype()
        arrs = [y, *xs] + ([weights] if weights is not None else [])
        if any(_is_torch(a) for a in arrs):
            import torch

            tdt = torch.float64 if dt == np.float64 else torch.float32
            keep = []
            for a in arrs:
                if not _is_torch(a):
                    a = torch.as_tensor(np.asarray(a))
                if not a.is_cuda:
                    a = a.cuda()
                keep.append(a.to(tdt).contiguous())  # cast like series_to_slice_inner (src/utils/mod.rs:134-205)
            self.space = _lib.PDS_DEVICE
            ptrs = [int(a.data_ptr()) for a in keep]
            self.n_rows = int(keep[0].shape[0])
            self.device_index = keep[0].device.index or 0
            self.torch_device = keep[0].device
        else:
            keep = [np.ascontiguousarray(np.asarray(a), dtype=dt) for a in arrs]
            self.space = _lib.PDS_HOST
            ptrs = [int(a.ctypes.data) for a in keep]
            self.n_rows = int(keep[0].shape[0])
            self.device_index = None
        for a in keep:
            if a.ndim != 1 or int(a.shape[0]) != self.n_rows:
                raise ValueError("all columns must be 1-D and of equal length")
        self.keep = keep
        self.n_feat = len(xs)
        nb = 1 + self.n_feat
        self.cols = (C.c_void_p * nb)(*ptrs[:nb])
        self.weights = C.c_void_p(ptrs[nb]) if weights is not None else C.c_void_p(None)


class Context:
    """One pds_ctx: a device, a stream, HBM workspace.  Not thread-safe; use one per thread."""

    def __init__(self, device: int = 0, stream=None):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        _lib.check(self._lib.pds_ctx_create(int(device), C.byref(self._h)))
        self.device = int(device)
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream) -> None:
        """stream: a torch.cuda.Stream, a raw hipStream_t integer, or None / 0 (HIP's default stream)."""
        raw = getattr(stream, "cuda_stream", stream)
        _lib.check(self._lib.pds_ctx_set_stream(self._h, C.c_void_p(int(raw) if raw else None)))
        self._stream_raw = int(raw) if raw else 0

    def follow_torch_stream(self, device) -> None:
        """
        Device-resident inputs are produced by work queued on torch's current stream: run on that stream
        so the kernels are ordered after the producers (and torch consumers after the kernels).
        """
        import torch

        raw = int(torch.cuda.current_stream(device).cuda_stream)
        if getattr(self, "_stream_raw", -1) != raw:
            self.set_stream(raw)
            self._stream_raw = raw

    def synchronize(self) -> None:
        _lib.check(self._lib.pds_ctx_synchronize(self._h))

    def set_option(self, name: str, value) -> None:
        """Behaviour switches of the context (include/pds_lstsq.h, pds_ctx_set_option): "keyed_sort", "wide_f32_native"; the
        defaults came from PDS_KEYED_SORT / PDS_WIDE_F32_NATIVE when the context was created."""
        _lib.check(self._lib.pds_ctx_set_option(self._h, str(name).encode(), C.c_longlong(int(value))))

    @property
    def num_cus(self) -> int:
        return int(self._lib.pds_ctx_num_cus(self._h))

    def signal_post(self, word_addr: int, value: int) -> None:
        """Stream ordered: store `value` (system scope) into the 32-bit word at `word_addr` behind everything queued so far."""
        _lib.check(self._lib.pds_signal_post(self._h, C.c_void_p(int(word_addr)), C.c_uint(int(value) & 0xFFFFFFFF)))

    def signal_wait(self, words_addr: int, n_words: int, value: int, timeout_ms: int = 2000) -> None:
        """Stream ordered: what follows on the stream runs once the `n_words` words at `words_addr` are all >= `value`."""
        _lib.check(self._lib.pds_signal_wait(self._h, C.c_void_p(int(words_addr)), int(n_words), C.c_uint(int(value) & 0xFFFFFFFF), int(timeout_ms)))

    def signal_wait_timeouts(self) -> int:
        n = C.c_int(0)
        _lib.check(self._lib.pds_signal_wait_status(self._h, C.byref(n)))
        return int(n.value)

    WORKSPACES = ("scratch", "stage", "solve", "keyed", "wkeyed")

    def workspace_bytes(self, which: str = "keyed") -> int:
        """Bytes held in one of the context's grow-only HBM workspaces (pds_ctx_workspace_bytes)."""
        return int(self._lib.pds_ctx_workspace_bytes(self._h, self.WORKSPACES.index(which)))

    KINDS = ("moments", "grouped_moments", "solve", "pass2", "rolling", "iterative")

    def set_timing(self, enable: bool) -> None:
        _lib.check(self._lib.pds_ctx_set_timing(self._h, int(bool(enable))))

    def get_timing(self, reset: bool = True) -> dict:
        """Per kernel class: (summed ms, launches) measured with HIP events on the context's stream."""
        ms = (C.c_double * 8)()
        cnt = (C.c_longlong * 8)()
        _lib.check(self._lib.pds_ctx_get_timing(self._h, ms, cnt, 8, int(bool(reset))))
        return {k: (ms[i], int(cnt[i])) for i, k in enumerate(self.KINDS)}

    def get_timing_samples(self, kind: str, reset: bool = True) -> list:
        """The individual launch durations (ms) of one kernel class since the last reset, oldest first."""
        buf = (C.c_double * 4096)()
        n = int(self._lib.pds_ctx_get_timing_samples(self._h, self.KINDS.index(kind), buf, 4096, int(bool(reset))))
        if n < 0:
            raise ValueError(kind)
        return [float(buf[i]) for i in range(n)]

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._lib.pds_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fn(self, name: str):
        return getattr(self._lib, name + _suffix())


_tls = threading.local()


def default_context() -> Context:
    """One context per calling thread (a pds_ctx is not thread-safe: stream, workspace and pinned staging are per call)."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is None:
        ctx = _tls.ctx = Context(0)
    return ctx


def _follow(ctx: "Context", cols: "_Cols") -> None:
    if cols.space == _lib.PDS_DEVICE:
        ctx.follow_torch_stream(cols.torch_device)


def _params(add_bias, l1_reg, l2_reg, tol, solver, positive, max_iter, singular_x_tol) -> _lib.LRParams:
    if singular_x_tol is None:  # dtype-aware default, expr_linear.py:184-186
        singular_x_tol = 1e-12 if config.LIN_REG_EXPR_F64 else 1e-6
    return _lib.LRParams(int(bool(add_bias)), float(l1_reg), float(l2_reg), float(tol), _lib.SOLVERS.get(solver, 0),
                         int(bool(positive)), int(max_iter), float(singular_x_tol))


def _out_like(cols: _Cols, shape):
    """An output buffer in the same memory space as the inputs."""
    if cols.space == _lib.PDS_DEVICE:
        import torch

        tdt = torch.float64 if config.LIN_REG_EXPR_F64 else torch.float32
        t = torch.empty(shape, dtype=tdt, device=cols.keep[0].device)
        return t, C.c_void_p(int(t.data_ptr()))
    a = np.empty(shape, dtype=_dtype())
    return a, C.c_void_p(a.ctypes.data)


def _out_u8(cols: _Cols, n):
    if cols.space == _lib.PDS_DEVICE:
        import torch

        t = torch.empty(n, dtype=torch.uint8, device=cols.keep[0].device)
        return t, C.c_void_p(int(t.data_ptr()))
    a = np.empty(n, dtype=np.uint8)
    return a, C.c_void_p(a.ctypes.data)


def parse_null_policy(value: str):
    """NullPolicy::try_from (src/linear/mod.rs:43-66): returns (policy code, fill value)."""
    v = str(value).lower()
    if v == "raise":
        return 0, 0.0
    if v == "skip":
        return 1, 0.0
    if v == "zero":
        return 2, 0.0
    if v == "one":
        return 2, 1.0
    if v == "ignore":
        return 3, 0.0
    try:
        return 2, float(value)
    except ValueError:
        raise ValueError("Invalid NullPolicy.") from None


def _is_arrow(a) -> bool:
    return type(a).__module__.startswith("pyarrow")


def _arrow_parts(a, dt):
    """(values ndarray, validity buffer address or 0, bit offset, keepalive) of a pyarrow (Chunked)Array."""
    import pyarrow as pa

    if isinstance(a, pa.ChunkedArray):
        a = a.combine_chunks() if a.num_chunks != 1 else a.chunk(0)  # like .rechunk()
    want = pa.float64() if dt == np.float64 else pa.float32()
    if a.type != want:
        a = a.cast(want)  # the Rust side casts non-matching inputs the same way (src/utils/mod.rs:134-205)
    vbuf, dbuf = a.buffers()
    vals = np.frombuffer(dbuf, dtype=dt)[a.offset : a.offset + len(a)]
    has_nulls = vbuf is not None and a.null_count > 0
    return vals, (vbuf.address if has_nulls else 0), a.offset, (a, vbuf, dbuf)


def _lin_reg_arrow(x, target, add_bias, return_pred, null_policy, prm, ctx):
    """pl_lr / pl_lr_pred on Arrow columns with validity bitmaps (host memory)."""
    dt = _dtype()
    parts = [_arrow_parts(c, dt) if _is_arrow(c) else (np.ascontiguousarray(np.asarray(c), dtype=dt), 0, 0, None)
             for c in (target, *x)]
    n = len(parts[0][0])
    if any(len(pt[0]) != n for pt in parts):
        raise ValueError("all columns must be 1-D and of equal length")
    nc = len(parts)
    cols = (C.c_void_p * nc)(*[pt[0].ctypes.data for pt in parts])
    vals = (C.c_void_p * nc)(*[pt[1] or None for pt in parts])
    offs = (C.c_int64 * nc)(*[pt[2] for pt in parts])
    code, fill = parse_null_policy(null_policy)
    pp = nc - 1 + int(bool(add_bias))
    coeffs = np.empty(pp, dtype=dt)
    is_null, n_used = C.c_int(0), C.c_int64(0)
    fillv = C.c_double(fill) if config.LIN_REG_EXPR_F64 else C.c_float(fill)
    if return_pred:
        pred, resid, valid = np.empty(n, dtype=dt), np.empty(n, dtype=dt), np.empty(n, dtype=np.uint8)
        pp_, rp_, vp_ = (C.c_void_p(a.ctypes.data) for a in (pred, resid, valid))
    else:
        pred = resid = valid = None
        pp_ = rp_ = vp_ = C.c_void_p(None)
    _lib.check(ctx.fn("pds_lr_nullable")(ctx._h, cols, vals, offs, nc - 1, C.c_int64(n), _lib.PDS_HOST, code, fillv, C.byref(prm),
                                         C.c_void_p(coeffs.ctypes.data), C.byref(is_null), pp_, rp_, vp_, C.byref(n_used)))
    if return_pred:
        if is_null.value:  # all-null {pred, resid} struct of the same length (linear_regression.rs:745-750)
            valid[:] = 0
            pred[:] = np.nan
            resid[:] = np.nan
        return pred, resid, valid.astype(bool)
    return None if is_null.value else coeffs


def lin_reg(*x, target, add_bias: bool = False, weights=None, return_pred: bool = False, l1_reg: float = 0.0,
            l2_reg: float = 0.0, tol: float = 1e-5, solver: str = "qr", max_iter: int = 200, positive: bool = False,
            singular_x_tol: float | None = None, null_policy: str = "skip", ctx: Context | None = None):
    """
    pds.lin_reg on null-free column buffers (pl_lr / pl_lr_pred).

    Returns the coefficient vector (bias last) or None when the rank gate fires (the reference returns
    a null list).  With return_pred=True returns (pred, resid) -- or None when gated (the reference
    returns an all-null struct).
    """
    if isinstance(target, (list, tuple)):  # multi-target: expr_linear.py:188-228
        if len(target) == 0:
            raise ValueError("If `target` is a list, it cannot be empty.")
        if len(target) == 1:
            return lin_reg(*x, target=target[0], add_bias=add_bias, weights=weights, return_pred=return_pred, l1_reg=l1_reg,
                           l2_reg=l2_reg, tol=tol, solver=solver, null_policy=null_policy, singular_x_tol=singular_x_tol, ctx=ctx)
        return _lin_reg_multi(x, list(target), add_bias, return_pred, l2_reg, solver, singular_x_tol, ctx or default_context())
    if max_iter <= 0:
        raise ValueError("Input `max_iter` must be a positive.")  # expr_linear.py:231-232
    ctx = ctx or default_context()
    if weights is None and any(_is_arrow(c) for c in (target, *x)):
        # Arrow columns may carry nulls: `null_policy` applies ("skip" is the reference's default, expr_linear.py:116);
        # with return_pred the result is (pred, resid, row_valid)
        prm = _params(add_bias, l1_reg, l2_reg, tol, solver, positive, max_iter, singular_x_tol)
        return _lin_reg_arrow(x, target, add_bias, return_pred, null_policy, prm, ctx)
    parse_null_policy(null_policy)  # validated even when no column can hold a null
    cols = _Cols(target, x, weights)
    _follow(ctx, cols)
    prm = _params(add_bias, l1_reg, l2_reg, tol, solver, positive, max_iter, singular_x_tol)
    pp = cols.n_feat + int(bool(add_bias))
    coeffs = np.empty(pp, dtype=_dtype())
    is_null = C.c_int(0)
    if return_pred:
        pred, pred_p = _out_like(cols, cols.n_rows)
        resid, resid_p = _out_like(cols, cols.n_rows)
        _lib.check(ctx.fn("pds_lr_pred")(ctx._h, cols.cols, cols.weights, cols.n_feat, C.c_int64(cols.n_rows), cols.space,
                                         C.byref(prm), C.c_void_p(coeffs.ctypes.data), C.byref(is_null), pred_p, resid_p))
        return None if is_null.value else (pred, resid)
    _lib.check(ctx.fn("pds_lr")(ctx._h, cols.cols, cols.weights, cols.n_feat, C.c_int64(cols.n_rows), cols.space,
                                C.byref(prm), C.c_void_p(coeffs.ctypes.data), C.byref(is_null)))
    return None if is_null.value else coeffs


def _lin_reg_multi(x, targets, add_bias, return_pred, l2_reg, solver, singular_x_tol, ctx):
    """pl_lr_multi / pl_lr_multi_pred: dict with the reference's struct field names (target_i[_pred|_resid])."""
    k = len(targets)
    cols = _Cols(targets[0], [*targets[1:], *x])  # C order: [t_0 .. t_{k-1}, x_1 .. x_p]
    _follow(ctx, cols)
    if singular_x_tol is None:
        singular_x_tol = 1e-12 if config.LIN_REG_EXPR_F64 else 1e-6
    p = len(x)
    pp = p + int(bool(add_bias))
    coeffs = np.empty((k, pp), dtype=_dtype())
    is_null = C.c_int(0)
    R = C.c_double if config.LIN_REG_EXPR_F64 else C.c_float
    if return_pred:
        pred, pr_p = _out_like(cols, (k, cols.n_rows))
        resid, rs_p = _out_like(cols, (k, cols.n_rows))
    else:
        pred = resid = None
        pr_p = rs_p = C.c_void_p(None)
    _lib.check(ctx.fn("pds_lr_multi")(ctx._h, cols.cols, k, p, C.c_int64(cols.n_rows), cols.space, int(bool(add_bias)),
                                      R(l2_reg), _lib.SOLVERS.get(solver, 0), R(singular_x_tol),
                                      C.c_void_p(coeffs.ctypes.data), C.byref(is_null), pr_p, rs_p))
    if return_pred:
        if is_null.value:
            return None
        out = {}
        for i in range(k):
            out[f"target_{i}_pred"] = pred[i]
            out[f"target_{i}_resid"] = resid[i]
        return out
    return {f"target_{i}": (None if is_null.value else coeffs[i]) for i in range(k)}


def query_ar_coeffs(x, lag: int, add_bias: bool = True, null_policy: str = "raise", ctx: Context | None = None):
    """
    Autoregressive coefficients of order `lag` (python/polars_ds/exprs/ts_features.py:419-461): `lin_reg` of x[t] on
    x[t-1] ... x[t-lag] over t = lag ... n-1, bias last.  The lagged features are *views* into the one series (the
    reference builds them with shift + slice): numpy / torch slices and pyarrow slices carry only a pointer and a bit
    offset, so no column is copied on the way to the Gram kernel, which reads element-aligned pointers.
    """
    if null_policy not in ("raise", "one", "zero"):
        import math

        try:
            z = float(null_policy)
            if not math.isfinite(z):
                raise ValueError
        except (TypeError, ValueError):
            raise ValueError("`null_polocy` must be 'raise', 'one', 'zero' or any finite numeric string for AR coefficients.") from None
    if lag <= 0:
        raise ValueError("`lag` must be > 0.")
    n = len(x)
    m = max(n - lag, 0)
    if _is_arrow(x):
        import pyarrow as pa

        if isinstance(x, pa.ChunkedArray):
            x = x.combine_chunks() if x.num_chunks != 1 else x.chunk(0)
        feats = [x.slice(lag - i, m) for i in range(1, lag + 1)]
        return lin_reg(*feats, target=x.slice(lag, m), add_bias=add_bias, null_policy=null_policy, ctx=ctx)
    feats = [x[lag - i : n - i] for i in range(1, lag + 1)]
    return lin_reg(*feats, target=x[lag:], add_bias=add_bias, null_policy=null_policy, ctx=ctx)


def lin_reg_w_rcond(*x, target, add_bias: bool = False, rcond: float = 0.0, l2_reg: float = 0.0, ctx: Context | None = None):
    """pds.lin_reg_w_rcond (pl_lr_w_rcond / pl_lr_w_rcond_f32): returns (coeffs, singular_values)."""
    ctx = ctx or default_context()
    cols = _Cols(target, x)
    _follow(ctx, cols)
    dt = _dtype()
    real = C.c_double if dt == np.float64 else C.c_float
    pp = cols.n_feat + int(bool(add_bias))
    # (kwargs.tol as T).max(T::EPSILON * max(nrows, p'))   linear_regression.rs:651-702, linear_regression_f32.rs:515-566
    rc = max(dt(rcond), np.finfo(dt).eps * dt(max(cols.n_rows, pp)))
    coeffs = np.empty(pp, dtype=dt)
    sv = np.empty(pp, dtype=dt)
    fn = ctx._lib.pds_lr_rcond_f64 if dt == np.float64 else ctx._lib.pds_lr_rcond_f32
    _lib.check(fn(ctx._h, cols.cols, cols.n_feat, C.c_int64(cols.n_rows), cols.space, int(bool(add_bias)), real(l2_reg), real(rc),
                  C.c_void_p(coeffs.ctypes.data), C.c_void_p(sv.ctypes.data)))
    return coeffs, sv


def elastic_net_fit(*x, target, l1_reg: float, l2_reg: float, add_bias: bool = False, tol: float = 1e-5, max_iter: int = 2000,
                    ctx: Context | None = None):
    """
    ElasticNet::fit_unchecked (src/linear/lr/lr_solvers.rs:139-164): always faer_coordinate_descent, also for l1_reg <= 0
    (pl_lr's dispatch would take the closed-form ridge there, which penalises with l2_reg where coordinate descent uses
    n_rows * l2_reg).  Returns the coefficients (bias last).
    """
    ctx = ctx or default_context()
    cols = _Cols(target, x)
    _follow(ctx, cols)
    dt = _dtype()
    prm = _lib.LRParams(int(bool(add_bias)), float(l1_reg), float(l2_reg), float(tol), 0, 0, int(max_iter), 0.0)
    pp = cols.n_feat + int(bool(add_bias))
    coeffs = np.empty(pp, dtype=dt)
    fn = ctx._lib.pds_elastic_net_f64 if dt == np.float64 else ctx._lib.pds_elastic_net_f32
    _lib.check(fn(ctx._h, cols.cols, cols.n_feat, C.c_int64(cols.n_rows), cols.space, C.byref(prm), C.c_void_p(coeffs.ctypes.data)))
    return coeffs


def lin_reg_report(*x, target, add_bias: bool = False, weights=None, std_err: str = "se", y_var: float | None = None,
                   feature_names: Sequence[str] | None = None, null_policy: str = "raise", ctx: Context | None = None) -> dict:
    """
    pds.lin_reg_report (pl_lin_reg_report / pl_wls_report).  `y_var` is what Polars evaluates as
    `target.var()` (ddof=1) and hands to the plugin as input 0 (expr_linear.py:614-617); when omitted it
    is taken from the moment pass (sum y, sum y^2 are by-products of the Gram build).
    Returns the 9 report columns as a dict of arrays (r2 / adj_r2 broadcast like the reference's struct).
    pyarrow columns may carry nulls: `null_policy` then applies as in series_to_mat_for_lr (the reference's default for
    this expression is "raise", expr_linear.py:566); `y_var` must be given in that case (Polars computes it on the
    original target).
    """
    ctx = ctx or default_context()
    arrow = weights is None and any(_is_arrow(c) for c in (target, *x))
    code, fill = parse_null_policy(null_policy)
    if arrow:
        if y_var is None:
            import pyarrow.compute as pc

            y_var = float(pc.variance(target, ddof=1).as_py())  # Polars' var() skips nulls the same way
        dt = _dtype()
        parts = [_arrow_parts(c, dt) if _is_arrow(c) else (np.ascontiguousarray(np.asarray(c), dtype=dt), 0, 0, None)
                 for c in (target, *x)]
        n = len(parts[0][0])
        if any(len(pt[0]) != n for pt in parts):
            raise ValueError("all columns must be 1-D and of equal length")
        nc = len(parts)
        n_feat = nc - 1
        pp = n_feat + int(bool(add_bias))
        outs = {k: np.empty(pp, dtype=dt) for k in ("beta", "std_err", "t", "p", "ci_lower", "ci_upper")}
        R = _lib.ReportF64 if config.LIN_REG_EXPR_F64 else _lib.ReportF32
        rep = R(*[C.c_void_p(outs[k].ctypes.data) for k in ("beta", "std_err", "t", "p", "ci_lower", "ci_upper")], 0.0, 0.0)
        cp = (C.c_void_p * nc)(*[pt[0].ctypes.data for pt in parts])
        vp = (C.c_void_p * nc)(*[pt[1] or None for pt in parts])
        op = (C.c_int64 * nc)(*[pt[2] for pt in parts])
        f = (C.c_double if config.LIN_REG_EXPR_F64 else C.c_float)
        n_used = C.c_int64(0)
        _lib.check(ctx.fn("pds_lin_reg_report_nullable")(ctx._h, cp, vp, op, n_feat, C.c_int64(n), _lib.PDS_HOST, code, f(fill),
                                                         int(bool(add_bias)), _lib.SE_TYPES.get(std_err, 0), f(y_var),
                                                         C.byref(rep), C.byref(n_used)))
        return _report_dict(outs, rep, n_feat, add_bias, std_err, False, feature_names, dt)
    cols = _Cols(target, x, weights)
    _follow(ctx, cols)
    pp = cols.n_feat + int(bool(add_bias))
    se_code = _lib.SE_TYPES.get(std_err, 0)
    if y_var is None:
        if weights is None:
            y_var = 0.0
            se_code |= _lib.PDS_REPORT_DERIVE_YVAR  # the library takes var(y) from its own pass over y (f64 sums)
        else:
            M = gram_moments(*x, target=target, ctx=ctx)
            q = cols.n_feat + 2
            n = float(cols.n_rows)
            sy, syy = float(M[q - 2, q - 1]), float(M[q - 1, q - 1])
            y_var = (syy - sy * sy / n) / (n - 1.0)
    dt = _dtype()
    outs = {k: np.empty(pp, dtype=dt) for k in ("beta", "std_err", "t", "p", "ci_lower", "ci_upper")}
    R = _lib.ReportF64 if config.LIN_REG_EXPR_F64 else _lib.ReportF32
    rep = R(*[C.c_void_p(outs[k].ctypes.data) for k in ("beta", "std_err", "t", "p", "ci_lower", "ci_upper")], 0.0, 0.0)
    yv = C.c_double(y_var) if config.LIN_REG_EXPR_F64 else C.c_float(y_var)
    _lib.check(ctx.fn("pds_lin_reg_report")(ctx._h, cols.cols, cols.weights, cols.n_feat, C.c_int64(cols.n_rows),
                                            cols.space, int(bool(add_bias)), se_code, yv,
                                            C.byref(rep)))
    return _report_dict(outs, rep, cols.n_feat, add_bias, std_err, weights is not None, feature_names, dt)


# ---- the stages of lin_reg_report as separate calls (row-sharded multi-GPU form: parallel.lin_reg_report_row_sharded)
def report_fit_from_moments(moments, *, add_bias: bool = False, ctx: Context | None = None):
    """Stage 2: (all-reduced) moment matrix -> (beta [p'], inv [p' x p'] = (X'X)^-1 by column-pivoted QR).  Host arrays."""
    ctx = ctx or default_context()
    dt = _dtype()
    if _is_torch(moments):
        moments = moments.detach().cpu().numpy()
    m = np.asfortranarray(np.asarray(moments, dtype=dt))
    q = int(m.shape[0])
    p = q - 2
    pp = p + int(bool(add_bias))
    beta = np.empty(pp, dtype=dt)
    inv = np.empty((pp, pp), dtype=dt, order="F")
    _lib.check(ctx.fn("pds_report_fit_from_moments")(ctx._h, C.c_void_p(m.ctypes.data), p, int(bool(add_bias)),
                                                     C.c_void_p(beta.ctypes.data), C.c_void_p(inv.ctypes.data)))
    return beta, inv


def report_partials(*x, target, beta, inv, add_bias: bool = False, weights=None, std_err: str = "se", ctx: Context | None = None):
    """Stage 3 on this rank's rows: [sum e^2, sum w e^2, meat block ((p+2)^2)] as one float64 vector (the all-reduce payload)."""
    ctx = ctx or default_context()
    cols = _Cols(target, x, weights)
    _follow(ctx, cols)
    dt = _dtype()
    q = cols.n_feat + 2
    beta = np.ascontiguousarray(beta, dtype=dt)
    inv = np.asfortranarray(np.asarray(inv, dtype=dt))
    out = np.zeros(2 + q * q, dtype=np.float64)
    _lib.check(ctx.fn("pds_report_partials")(ctx._h, cols.cols, cols.weights, cols.n_feat, C.c_int64(cols.n_rows), cols.space,
                                             int(bool(add_bias)), _lib.SE_TYPES.get(std_err, 0), C.c_void_p(beta.ctypes.data),
                                             C.c_void_p(inv.ctypes.data), C.c_void_p(out.ctypes.data)))
    return out


def report_finish(beta, inv, partials, *, n_rows_total: int, y_var: float, add_bias: bool = False, weighted: bool = False,
                  std_err: str = "se", feature_names: Sequence[str] | None = None) -> dict:
    """Stage 4: the O(p'^2) epilogue on the summed partials (no device work)."""
    lib = _lib.load()
    dt = _dtype()
    beta = np.ascontiguousarray(beta, dtype=dt)
    inv = np.asfortranarray(np.asarray(inv, dtype=dt))
    partials = np.ascontiguousarray(partials, dtype=np.float64)
    pp = int(beta.shape[0])
    n_feat = pp - int(bool(add_bias))
    outs = {k: np.empty(pp, dtype=dt) for k in ("beta", "std_err", "t", "p", "ci_lower", "ci_upper")}
    R = _lib.ReportF64 if config.LIN_REG_EXPR_F64 else _lib.ReportF32
    rep = R(*[C.c_void_p(outs[k].ctypes.data) for k in ("beta", "std_err", "t", "p", "ci_lower", "ci_upper")], 0.0, 0.0)
    yv = C.c_double(y_var) if config.LIN_REG_EXPR_F64 else C.c_float(y_var)
    fn = getattr(lib, "pds_report_finish" + _suffix())
    _lib.check(fn(n_feat, int(bool(add_bias)), _lib.SE_TYPES.get(std_err, 0), int(bool(weighted)), C.c_int64(int(n_rows_total)), yv,
                  C.c_void_p(beta.ctypes.data), C.c_void_p(inv.ctypes.data), C.c_void_p(partials.ctypes.data), C.byref(rep)))
    return _report_dict(outs, rep, n_feat, add_bias, std_err, weighted, feature_names, dt)


def _report_dict(outs, rep, n_feat, add_bias, std_err, weighted, feature_names, dt):
    pp = n_feat + int(bool(add_bias))
    names = list(feature_names) if feature_names is not None else [f"x{i + 1}" for i in range(n_feat)]
    if add_bias:
        names.append("__bias__")  # linear_regression.rs:843-845
    se_name = {"se": "std_err", "hc0": "hc0_se", "hc1": "hc1_se", "hc2": "hc2_se", "hc3": "hc3_se"}.get(std_err, "std_err")
    if weighted:
        se_name = "std_err"
    return {
        "features": names, "beta": outs["beta"], se_name: outs["std_err"], "t": outs["t"], "p>|t|": outs["p"],
        "0.025": outs["ci_lower"], "0.975": outs["ci_upper"],
        "r2": np.full(pp, rep.r2, dtype=dt), "adj_r2": np.full(pp, rep.adj_r2, dtype=dt),
    }


def lin_reg_by(*x, target, group_offsets, add_bias: bool = False, l1_reg: float = 0.0, l2_reg: float = 0.0,
               tol: float = 1e-5, solver: str = "qr", max_iter: int = 200, positive: bool = False,
               singular_x_tol: float | None = None, null_policy: str = "skip", weights=None, ctx: Context | None = None):
    """
    The key-aware batched form of `df.group_by(key).agg(pds.lin_reg(...))` (SURVEY.md 8b "pl_lr_by").
    Rows of a group are contiguous; group g = rows [group_offsets[g], group_offsets[g+1]).
    Returns (coeffs [n_groups, p'], is_null [n_groups]) in the memory space of the inputs.
    pyarrow columns may carry nulls; `null_policy` then applies inside every group, as it does when Polars calls pl_lr
    once per group.  `l1_reg` / `l2_reg` / `positive` select the method per group exactly as `lin_reg` does
    (lasso / elastic net / non-negative fits run the reference's coordinate descent on every group's Gram matrix).
    `weights`: per-group weighted least squares (faer_weighted_lr: no gate, no penalties, like `lin_reg(weights=...)`).
    """
    if max_iter <= 0:
        raise ValueError("Input `max_iter` must be a positive.")  # expr_linear.py:231-232
    ctx = ctx or default_context()
    if weights is not None:
        cols = _Cols(target, x, weights)
        _follow(ctx, cols)
        prm = _params(add_bias, 0.0, 0.0, tol, solver, False, max_iter, 0.0)
        pp = cols.n_feat + int(bool(add_bias))
        if cols.space == _lib.PDS_DEVICE:
            import torch

            off = group_offsets if _is_torch(group_offsets) else torch.as_tensor(np.asarray(group_offsets))
            off = off.to(device=cols.keep[0].device, dtype=torch.int64).contiguous()
            off_p = C.c_void_p(int(off.data_ptr()))
        else:
            off = np.ascontiguousarray(np.asarray(group_offsets), dtype=np.int64)
            off_p = C.c_void_p(off.ctypes.data)
        ng = int(off.shape[0]) - 1
        coeffs, co_p = _out_like(cols, (ng, pp))
        nulls, nu_p = _out_u8(cols, ng)
        _lib.check(ctx.fn("pds_lr_grouped_weighted")(ctx._h, cols.cols, cols.weights, cols.n_feat, C.c_int64(cols.n_rows), off_p,
                                                     C.c_int64(ng), cols.space, C.byref(prm), co_p, nu_p))
        return coeffs, nulls
    code, fill = parse_null_policy(null_policy)
    if any(_is_arrow(c) for c in (target, *x)):
        dt = _dtype()
        parts = [_arrow_parts(c, dt) if _is_arrow(c) else (np.ascontiguousarray(np.asarray(c), dtype=dt), 0, 0, None)
                 for c in (target, *x)]
        n = len(parts[0][0])
        if any(len(pt[0]) != n for pt in parts):
            raise ValueError("all columns must be 1-D and of equal length")
        nc = len(parts)
        prm = _params(add_bias, l1_reg, l2_reg, tol, solver, positive, max_iter, singular_x_tol)
        pp = nc - 1 + int(bool(add_bias))
        off = np.ascontiguousarray(np.asarray(group_offsets), dtype=np.int64)
        ng = int(off.shape[0]) - 1
        coeffs = np.empty((ng, pp), dtype=dt)
        nulls = np.empty(ng, dtype=np.uint8)
        cp = (C.c_void_p * nc)(*[pt[0].ctypes.data for pt in parts])
        vp = (C.c_void_p * nc)(*[pt[1] or None for pt in parts])
        op = (C.c_int64 * nc)(*[pt[2] for pt in parts])
        f = (C.c_double if config.LIN_REG_EXPR_F64 else C.c_float)
        _lib.check(ctx.fn("pds_lr_grouped_nullable")(ctx._h, cp, vp, op, nc - 1, C.c_int64(n), C.c_void_p(off.ctypes.data),
                                                     C.c_int64(ng), _lib.PDS_HOST, code, f(fill), C.byref(prm),
                                                     C.c_void_p(coeffs.ctypes.data), C.c_void_p(nulls.ctypes.data)))
        return coeffs, nulls
    cols = _Cols(target, x)
    _follow(ctx, cols)
    prm = _params(add_bias, l1_reg, l2_reg, tol, solver, positive, max_iter, singular_x_tol)
    pp = cols.n_feat + int(bool(add_bias))
    if cols.space == _lib.PDS_DEVICE:
        import torch

        off = group_offsets if _is_torch(group_offsets) else torch.as_tensor(np.asarray(group_offsets))
        off = off.to(device=cols.keep[0].device, dtype=torch.int64).contiguous()
        off_p = C.c_void_p(int(off.data_ptr()))
    else:
        off = np.ascontiguousarray(np.asarray(group_offsets), dtype=np.int64)
        off_p = C.c_void_p(off.ctypes.data)
    ng = int(off.shape[0]) - 1
    coeffs, co_p = _out_like(cols, (ng, pp))
    nulls, nu_p = _out_u8(cols, ng)
    _lib.check(ctx.fn("pds_lr_grouped")(ctx._h, cols.cols, cols.n_feat, C.c_int64(cols.n_rows), off_p, C.c_int64(ng),
                                        cols.space, C.byref(prm), co_p, nu_p))
    return coeffs, nulls


class DeviceBlock:
    """
    `bytes` of device memory that kernels of ANOTHER process can write (include/pds_lstsq.h: pds_device_alloc / pds_ipc_export /
    pds_ipc_open): the owner allocates and exports 64 opaque handle bytes; a peer maps them with `DeviceBlock.open(handle, device)`
    and uses `.addr` (+ byte offsets) as a raw output address.  `tensor(dtype, shape, byte_offset)` views the OWNER's block as a torch
    tensor (`__cuda_array_interface__`).  Freed / unmapped by `close()` (or at garbage collection).
    """

    def __init__(self, nbytes: int, device: int, fine_grained: bool = True):
        self._lib = _lib.load()
        self.device, self.nbytes, self.owner = int(device), int(nbytes), True
        p = C.c_void_p()
        # (fine grained: coherent across devices -- rows another device's kernel writes must not sit stale in this device's L2)
        _lib.check(self._lib.pds_device_alloc(self.device, C.c_size_t(self.nbytes), int(bool(fine_grained)), C.byref(p)))
        self.addr = int(p.value)

    @classmethod
    def open(cls, handle: bytes, device: int, nbytes: int = 0):
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self.device, self.nbytes, self.owner = int(device), int(nbytes), False
        p = C.c_void_p()
        hb = (C.c_ubyte * 64).from_buffer_copy(bytes(handle))
        _lib.check(self._lib.pds_ipc_open(self.device, hb, C.byref(p)))
        self.addr = int(p.value)
        return self

    def handle(self) -> bytes:
        hb = (C.c_ubyte * 64)()
        _lib.check(self._lib.pds_ipc_export(self.device, C.c_void_p(self.addr), hb))
        return bytes(hb)

    def tensor(self, dtype, shape, byte_offset: int = 0):
        import torch

        class _View:  # (torch keeps the object alive as the tensor's base)
            pass

        v = _View()
        v.block = self
        typestr = {torch.float64: "<f8", torch.float32: "<f4", torch.uint8: "|u1", torch.int64: "<i8"}[dtype]
        v.__cuda_array_interface__ = {"shape": tuple(int(d) for d in shape), "typestr": typestr, "data": (self.addr + int(byte_offset), False),
                                      "version": 2, "strides": None}
        return torch.as_tensor(v, device=torch.device("cuda", self.device))

    def close(self) -> None:
        if getattr(self, "addr", 0):
            if self.owner:
                self._lib.pds_device_free(self.device, C.c_void_p(self.addr))
            else:
                self._lib.pds_ipc_close(self.device, C.c_void_p(self.addr))
            self.addr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupedFit:
    """
    A prepared `lin_reg_by` over DEVICE-resident columns with contiguous groups: the column pointer table, the parameter block, the
    device offsets and the result buffers are set up once; `run()` is then one C call (`pds_lr_grouped_*`) and nothing else -- no
    allocation, no dtype / contiguity checks, no Python per column.  What a host engine that evaluates the same expression over a
    resident frame repeatedly (or one rank of the group-sharded step, parallel.GroupedShardPlan) holds on to.

    `out` / `out_null`: where the coefficients ([n_groups, p'] of the config dtype, contiguous) and the null flags ([n_groups] uint8)
    go -- e.g. rows of a larger gathered result, so that the fit writes them in place.  By default they are allocated here, once.
    Every `run()` recomputes and overwrites them; it returns the same two tensors.
    `out_addr` / `out_null_addr` (both or neither; instead of `out` / `out_null`): raw device addresses of such rows in memory that
    is not a torch tensor of this process -- the assembled result of ANOTHER rank, mapped with `DeviceBlock.open` (the direct gather of
    parallel.GroupedShardPlan): the kernel's stores cross the link.  `run()` then returns (None, None).
    """

    def __init__(self, *x, target, group_offsets, add_bias: bool = False, l1_reg: float = 0.0, l2_reg: float = 0.0, tol: float = 1e-5,
                 solver: str = "qr", max_iter: int = 200, positive: bool = False, singular_x_tol: float | None = None, out=None,
                 out_null=None, out_addr: int | None = None, out_null_addr: int | None = None, ctx: Context | None = None):
        if max_iter <= 0:
            raise ValueError("Input `max_iter` must be a positive.")  # expr_linear.py:231-232
        import torch

        self.ctx = ctx or default_context()
        cols = _Cols(target, x)
        if cols.space != _lib.PDS_DEVICE:
            raise ValueError("GroupedFit prepares device-resident frames (host frames: lin_reg_by)")
        self._cols = cols
        self._prm = _params(add_bias, l1_reg, l2_reg, tol, solver, positive, max_iter, singular_x_tol)
        self._off, off_p = _offsets_arg(cols, group_offsets)
        ng = int(self._off.shape[0]) - 1
        pp = cols.n_feat + int(bool(add_bias))
        tdt = torch.float64 if config.LIN_REG_EXPR_F64 else torch.float32
        dev = cols.keep[0].device
        if (out_addr is None) != (out_null_addr is None) or (out_addr is not None and (out is not None or out_null is not None)):
            raise ValueError("`out_addr` and `out_null_addr` come together, instead of `out` / `out_null`")
        if out_addr is not None:
            self.coeffs = self.is_null = None
            self.n_groups = ng
            self._fn = self.ctx.fn("pds_lr_grouped")
            self._args = (self.ctx._h, cols.cols, cols.n_feat, C.c_int64(cols.n_rows), off_p, C.c_int64(ng), cols.space, C.byref(self._prm),
                          C.c_void_p(int(out_addr)), C.c_void_p(int(out_null_addr)))
            self._dev = cols.torch_device
            return
        if out is None:
            out = torch.empty((ng, pp), dtype=tdt, device=dev)
        if out_null is None:
            out_null = torch.empty((ng,), dtype=torch.uint8, device=dev)
        if tuple(out.shape) != (ng, pp) or out.dtype != tdt or out.device != dev or not out.is_contiguous():
            raise ValueError(f"`out` must be a contiguous [{ng}, {pp}] {tdt} tensor on {dev}")
        if tuple(out_null.shape) != (ng,) or out_null.dtype != torch.uint8 or out_null.device != dev or not out_null.is_contiguous():
            raise ValueError(f"`out_null` must be a contiguous [{ng}] uint8 tensor on {dev}")
        self.coeffs, self.is_null = out, out_null
        self.n_groups = ng
        self._fn = self.ctx.fn("pds_lr_grouped")
        self._args = (self.ctx._h, cols.cols, cols.n_feat, C.c_int64(cols.n_rows), off_p, C.c_int64(ng), cols.space, C.byref(self._prm),
                      C.c_void_p(int(out.data_ptr())), C.c_void_p(int(out_null.data_ptr())))
        self._dev = cols.torch_device

    def run(self):
        if self.n_groups <= 0:
            return self.coeffs, self.is_null
        self.ctx.follow_torch_stream(self._dev)
        rc = self._fn(*self._args)
        if rc:
            _lib.check(rc)
        return self.coeffs, self.is_null  # ((None, None) with raw output addresses)


def lin_reg_by_key(*x, target, key, add_bias: bool = False, l1_reg: float = 0.0, l2_reg: float = 0.0, tol: float = 1e-5,
                   solver: str = "qr", max_iter: int = 200, positive: bool = False, singular_x_tol: float | None = None,
                   max_groups: int | None = None, ctx: Context | None = None):
    """
    `df.group_by(key).agg(pds.lin_reg(...))` for an integer key column in ANY row order: the frame is brought into key
    order on the device (radix sort of (key, row) pairs + column gather; nothing moves when the keys are already
    non-decreasing), then every group is fitted as in `lin_reg_by`.
    Returns (keys [n_groups] ascending, coeffs [n_groups, p'], is_null [n_groups]) in the memory space of the inputs.
    """
    if max_iter <= 0:
        raise ValueError("Input `max_iter` must be a positive.")  # expr_linear.py:231-232
    ctx = ctx or default_context()
    cols = _Cols(target, x)
    _follow(ctx, cols)
    prm = _params(add_bias, l1_reg, l2_reg, tol, solver, positive, max_iter, singular_x_tol)
    pp = cols.n_feat + int(bool(add_bias))
    # outputs are sized for max_groups distinct keys; without a hint start from n_rows / 16 (at least 2^20) and repeat with
    # the exact count when there are more -- sizing for one key per row would allocate p' x n_rows coefficients
    n_rows = cols.n_rows
    cap = int(max_groups) if max_groups is not None else (n_rows if n_rows <= (1 << 20) else max(1 << 20, n_rows // 16))
    if cols.space == _lib.PDS_DEVICE:
        import torch

        k = key if _is_torch(key) else torch.as_tensor(np.asarray(key))
        k = k.to(device=cols.keep[0].device, dtype=torch.int64).contiguous()
        k_p = C.c_void_p(int(k.data_ptr()))
    else:
        k = np.ascontiguousarray(np.asarray(key), dtype=np.int64)
        k_p = C.c_void_p(k.ctypes.data)
    if int(k.shape[0]) != n_rows:
        raise ValueError("`key` must have one entry per row")
    ng = C.c_int64(0)
    while True:
        if cols.space == _lib.PDS_DEVICE:
            ok = torch.empty(cap, dtype=torch.int64, device=k.device)
            ok_p = C.c_void_p(int(ok.data_ptr()))
        else:
            ok = np.empty(cap, dtype=np.int64)
            ok_p = C.c_void_p(ok.ctypes.data)
        coeffs, co_p = _out_like(cols, (cap, pp))
        nulls, nu_p = _out_u8(cols, cap)
        rc = ctx.fn("pds_lr_by_key")(ctx._h, cols.cols, k_p, cols.n_feat, C.c_int64(n_rows), cols.space, C.byref(prm),
                                     C.c_int64(cap), ok_p, co_p, nu_p, C.byref(ng))
        if rc != 0 and max_groups is None and int(ng.value) > cap:
            cap = int(ng.value)  # more distinct keys than the first guess
            continue
        _lib.check(rc)
        break
    g = int(ng.value)
    return ok[:g], coeffs[:g], nulls[:g]


def lin_reg_by_key_multi(*x, target, key, contexts, n_slices: int = 0, add_bias: bool = False, l1_reg: float = 0.0, l2_reg: float = 0.0,
                         tol: float = 1e-5, solver: str = "qr", max_iter: int = 200, positive: bool = False,
                         singular_x_tol: float | None = None, max_groups: int | None = None):
    """
    `lin_reg_by_key` for a HOST frame whose keys are in order, driven through several contexts from this one process
    (pds_lr_by_key_multi_*): the frame is cut into row slices at group boundaries and slice s is fitted by contexts[s mod n] on
    that context's device -- contexts on different devices pull their shards over their own PCIe links,
    several contexts on one device overlap a slice's transfer with the previous slice's fit.  Same results as `lin_reg_by_key`.
    Returns (keys, coeffs, is_null) as numpy arrays.
    """
    if max_iter <= 0:
        raise ValueError("Input `max_iter` must be a positive.")
    cols = _Cols(target, x)
    if cols.space != _lib.PDS_HOST:
        raise ValueError("lin_reg_by_key_multi takes host-resident columns (numpy)")
    prm = _params(add_bias, l1_reg, l2_reg, tol, solver, positive, max_iter, singular_x_tol)
    pp = cols.n_feat + int(bool(add_bias))
    n_rows = cols.n_rows
    k = np.ascontiguousarray(np.asarray(key), dtype=np.int64)
    if int(k.shape[0]) != n_rows:
        raise ValueError("`key` must have one entry per row")
    cap = int(max_groups) if max_groups is not None else (n_rows if n_rows <= (1 << 20) else max(1 << 20, n_rows // 16))
    handles = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    ng = C.c_int64(0)
    fn = contexts[0].fn("pds_lr_by_key_multi")
    while True:
        ok = np.empty(cap, dtype=np.int64)
        coeffs = np.empty((cap, pp), dtype=_dtype())
        nulls = np.empty(cap, dtype=np.uint8)
        rc = fn(handles, len(contexts), int(n_slices), cols.cols, C.c_void_p(k.ctypes.data), cols.n_feat, C.c_int64(n_rows), C.byref(prm),
                C.c_int64(cap), C.c_void_p(ok.ctypes.data), C.c_void_p(coeffs.ctypes.data), C.c_void_p(nulls.ctypes.data), C.byref(ng))
        if rc != 0 and max_groups is None and int(ng.value) > cap:
            cap = int(ng.value)
            continue
        _lib.check(rc)
        break
    g = int(ng.value)
    return ok[:g], coeffs[:g], nulls[:g]


# ---- the exchange steps of SURVEY.md 8(e) between the contexts of ONE process (include/pds_lstsq.h: pds_allreduce_sum_*,
# pds_scatter_rows_*, pds_gather_*): what a host without a collective library composes the multi-device paths from
def _exchange_suffix(t) -> str:
    import torch

    if t.dtype == torch.float64:
        return "_f64"
    if t.dtype == torch.float32:
        return "_f32"
    raise ValueError("exchange steps move float64 / float32 device tensors")


def allreduce_sum(contexts, tensors, prefix: bool = False) -> None:
    """tensors[c]: a contiguous CUDA tensor on contexts[c]'s device, all of one shape and dtype.  In place: every tensor becomes the
    element-wise sum (prefix=True: tensor c becomes the sum of tensors 0 .. c-1 -- the exclusive scan of the row-sharded expanding fit)."""
    n = len(contexts)
    if len(tensors) != n or n < 1:
        raise ValueError("one tensor per context")
    suf = _exchange_suffix(tensors[0])
    for x in tensors:
        if not (x.is_cuda and x.is_contiguous() and x.shape == tensors[0].shape and x.dtype == tensors[0].dtype):
            raise ValueError("contiguous CUDA tensors of one shape and dtype")
    h = (C.c_void_p * n)(*[c._h for c in contexts])
    b = (C.c_void_p * n)(*[x.data_ptr() for x in tensors])
    _lib.check(getattr(_lib.load(), "pds_allreduce_sum" + suf)(h, n, b, C.c_int64(tensors[0].numel()), 1 if prefix else 0))


def scatter_rows(contexts, cols, bounds):
    """cols: CUDA column tensors on contexts[0]'s device; rows [bounds[c], bounds[c+1]) of every column -> new tensors on contexts[c]'s
    device.  Returns a list (per context) of lists of column tensors; rank 0 gets views of the frame itself."""
    import torch

    n = len(contexts)
    suf = _exchange_suffix(cols[0])
    bounds = [int(v) for v in bounds]
    out = [[c[bounds[0]:bounds[1]] for c in cols]]
    for r in range(1, n):
        out.append([torch.empty(bounds[r + 1] - bounds[r], dtype=cols[0].dtype, device=torch.device("cuda", contexts[r].device)) for _ in cols])
    h = (C.c_void_p * n)(*[c._h for c in contexts])
    src = (C.c_void_p * len(cols))(*[c.data_ptr() for c in cols])
    tabs = [(C.c_void_p * len(cols))(*[x.data_ptr() for x in o]) for o in out]
    dst = (C.c_void_p * n)(*([None] + [C.cast(tb, C.c_void_p) for tb in tabs[1:]]))
    bd = (C.c_int64 * (n + 1))(*bounds)
    _lib.check(getattr(_lib.load(), "pds_scatter_rows" + suf)(h, n, src, len(cols), bd, dst))
    return out


def gather(contexts, blocks):
    """blocks[c]: a contiguous CUDA tensor on contexts[c]'s device -> one tensor on contexts[0]'s device, the blocks' elements back
    to back in rank order."""
    import torch

    n = len(contexts)
    suf = _exchange_suffix(blocks[0])
    counts = [int(b.numel()) for b in blocks]
    out = torch.empty(sum(counts), dtype=blocks[0].dtype, device=torch.device("cuda", contexts[0].device))
    h = (C.c_void_p * n)(*[c._h for c in contexts])
    src = (C.c_void_p * n)(*[b.data_ptr() for b in blocks])
    cn = (C.c_int64 * n)(*counts)
    _lib.check(getattr(_lib.load(), "pds_gather" + suf)(h, n, src, cn, C.c_void_p(out.data_ptr())))
    return out


def _offsets_arg(cols: "_Cols", group_offsets):
    if cols.space == _lib.PDS_DEVICE:
        import torch

        off = group_offsets if _is_torch(group_offsets) else torch.as_tensor(np.asarray(group_offsets))
        off = off.to(device=cols.keep[0].device, dtype=torch.int64).contiguous()
        return off, C.c_void_p(int(off.data_ptr()))
    off = np.ascontiguousarray(np.asarray(group_offsets), dtype=np.int64)
    return off, C.c_void_p(off.ctypes.data)


def lin_reg_by_pred(*x, target, group_offsets, add_bias: bool = False, l1_reg: float = 0.0, l2_reg: float = 0.0, tol: float = 1e-5,
                    solver: str = "qr", max_iter: int = 200, positive: bool = False, singular_x_tol: float | None = None,
                    weights=None, ctx: Context | None = None):
    """
    `df.group_by(key).agg(pds.lin_reg(..., return_pred=True))` / `pds.lin_reg(..., return_pred=True).over(key)` for contiguous
    groups (tests/test_linear_exprs.py:435-474): every group is fitted as `lin_reg_by` fits it, then one more pass writes
    pred = x . beta_group and resid = y - pred for every row.
    Returns (pred [n_rows], resid [n_rows], row_is_null [n_rows], coeffs [n_groups, p'], group_is_null [n_groups]); rows of a
    null group (gated, or fewer rows than coefficients) are NaN / flagged -- the reference returns an all-null struct there.
    """
    if max_iter <= 0:
        raise ValueError("Input `max_iter` must be a positive.")
    ctx = ctx or default_context()
    cols = _Cols(target, x, weights)
    _follow(ctx, cols)
    prm = (_params(add_bias, 0.0, 0.0, tol, solver, False, max_iter, 0.0) if weights is not None
           else _params(add_bias, l1_reg, l2_reg, tol, solver, positive, max_iter, singular_x_tol))
    pp = cols.n_feat + int(bool(add_bias))
    off, off_p = _offsets_arg(cols, group_offsets)
    ng = int(off.shape[0]) - 1
    coeffs, co_p = _out_like(cols, (ng, pp))
    nulls, nu_p = _out_u8(cols, ng)
    pred, pr_p = _out_like(cols, cols.n_rows)
    resid, re_p = _out_like(cols, cols.n_rows)
    rnull, rn_p = _out_u8(cols, cols.n_rows)
    _lib.check(ctx.fn("pds_lr_grouped_pred")(ctx._h, cols.cols, cols.weights, cols.n_feat, C.c_int64(cols.n_rows), off_p, C.c_int64(ng),
                                             cols.space, C.byref(prm), co_p, nu_p, pr_p, re_p, rn_p))
    return pred, resid, rnull, coeffs, nulls


def lin_reg_by_key_pred(*x, target, key, add_bias: bool = False, l1_reg: float = 0.0, l2_reg: float = 0.0, tol: float = 1e-5,
                        solver: str = "qr", max_iter: int = 200, positive: bool = False, singular_x_tol: float | None = None,
                        weights=None, ctx: Context | None = None):
    """
    The same for an integer key column in ANY row order: pred / resid come back in the FRAME's row order (the device orders the
    frame by key, fits, and sends every row's prediction back to where the row is).  Returns (pred, resid, row_is_null).
    """
    if max_iter <= 0:
        raise ValueError("Input `max_iter` must be a positive.")
    ctx = ctx or default_context()
    cols = _Cols(target, x, weights)
    _follow(ctx, cols)
    prm = (_params(add_bias, 0.0, 0.0, tol, solver, False, max_iter, 0.0) if weights is not None
           else _params(add_bias, l1_reg, l2_reg, tol, solver, positive, max_iter, singular_x_tol))
    n_rows = cols.n_rows
    if cols.space == _lib.PDS_DEVICE:
        import torch

        k = key if _is_torch(key) else torch.as_tensor(np.asarray(key))
        k = k.to(device=cols.keep[0].device, dtype=torch.int64).contiguous()
        k_p = C.c_void_p(int(k.data_ptr()))
    else:
        k = np.ascontiguousarray(np.asarray(key), dtype=np.int64)
        k_p = C.c_void_p(k.ctypes.data)
    if int(k.shape[0]) != n_rows:
        raise ValueError("`key` must have one entry per row")
    pred, pr_p = _out_like(cols, n_rows)
    resid, re_p = _out_like(cols, n_rows)
    rnull, rn_p = _out_u8(cols, n_rows)
    _lib.check(ctx.fn("pds_lr_by_key_pred")(ctx._h, cols.cols, cols.weights, k_p, cols.n_feat, C.c_int64(n_rows), cols.space,
                                            C.byref(prm), C.c_int64(n_rows), None, None, None, None, pr_p, re_p, rn_p))
    return pred, resid, rnull


def lin_reg_by_key_pred_multi(*x, target, key, contexts, n_slices: int = 0, add_bias: bool = False, l1_reg: float = 0.0, l2_reg: float = 0.0,
                              tol: float = 1e-5, solver: str = "qr", max_iter: int = 200, positive: bool = False,
                              singular_x_tol: float | None = None, weights=None):
    """
    `lin_reg_by_key_pred` for a HOST frame whose keys are in order, over several contexts of this one process
    (pds_lr_by_key_pred_multi_*): independent row slices cut at group boundaries, slice s on contexts[s mod n]; a slice's
    predictions travel back while the next slice's rows arrive.  Returns (pred, resid, row_is_null) as numpy arrays, frame order.
    """
    if max_iter <= 0:
        raise ValueError("Input `max_iter` must be a positive.")
    cols = _Cols(target, x, weights)
    if cols.space != _lib.PDS_HOST:
        raise ValueError("lin_reg_by_key_pred_multi takes host-resident columns (numpy)")
    prm = (_params(add_bias, 0.0, 0.0, tol, solver, False, max_iter, 0.0) if weights is not None
           else _params(add_bias, l1_reg, l2_reg, tol, solver, positive, max_iter, singular_x_tol))
    n_rows = cols.n_rows
    k = np.ascontiguousarray(np.asarray(key), dtype=np.int64)
    if int(k.shape[0]) != n_rows:
        raise ValueError("`key` must have one entry per row")
    pred, resid = np.empty(n_rows, dtype=_dtype()), np.empty(n_rows, dtype=_dtype())
    rnull = np.empty(n_rows, dtype=np.uint8)
    handles = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    _lib.check(contexts[0].fn("pds_lr_by_key_pred_multi")(handles, len(contexts), int(n_slices), cols.cols, cols.weights,
                                                          C.c_void_p(k.ctypes.data), cols.n_feat, C.c_int64(n_rows), C.byref(prm),
                                                          C.c_void_p(pred.ctypes.data), C.c_void_p(resid.ctypes.data),
                                                          C.c_void_p(rnull.ctypes.data)))
    return pred, resid, rnull


def _windowed(name, x, target, n, add_bias, l2_reg, min_size, ctx, seed_moments=None):
    ctx = ctx or default_context()
    cols = _Cols(target, x)
    _follow(ctx, cols)
    pp = cols.n_feat + int(bool(add_bias))
    coeffs, co_p = _out_like(cols, (cols.n_rows, pp))
    pred, pr_p = _out_like(cols, cols.n_rows)
    valid, va_p = _out_u8(cols, cols.n_rows)
    lam = C.c_double(abs(l2_reg)) if config.LIN_REG_EXPR_F64 else C.c_float(abs(l2_reg))  # `lambda`: abs(l2_reg) :546-552
    if name == "pds_rolling_lr":
        _lib.check(ctx.fn(name)(ctx._h, cols.cols, cols.n_feat, C.c_int64(cols.n_rows), cols.space, int(bool(add_bias)),
                                C.c_int64(n), C.c_int64(min_size), lam, co_p, pr_p, va_p))
    elif seed_moments is not None:
        q = cols.n_feat + 2
        seed = np.asarray(seed_moments.cpu() if _is_torch(seed_moments) else seed_moments, dtype=_dtype())
        if seed.shape != (q, q):
            raise ValueError(f"seed_moments must be the ({q}, {q}) augmented moment matrix of the preceding rows")
        seed = np.asfortranarray(seed)
        _lib.check(ctx.fn("pds_recursive_lr_seeded")(ctx._h, cols.cols, cols.n_feat, C.c_int64(cols.n_rows), cols.space,
                                                     int(bool(add_bias)), C.c_int64(n), lam, C.c_void_p(seed.ctypes.data),
                                                     co_p, pr_p, va_p))
    else:
        _lib.check(ctx.fn(name)(ctx._h, cols.cols, cols.n_feat, C.c_int64(cols.n_rows), cols.space, int(bool(add_bias)),
                                C.c_int64(n), lam, co_p, pr_p, va_p))
    return coeffs, pred, valid


def rolling_lin_reg(*x, target, window_size: int, add_bias: bool = False, l2_reg: float = 0.0,
                    min_valid_rows: int | None = None, skip_non_finite: bool = False, ctx: Context | None = None):
    """
    pds.rolling_lin_reg (pl_rolling_lr).  Returns (coeffs [N, p'], pred [N], valid [N]); the first
    window_size-1 rows are invalid (null in the reference).  skip_non_finite=True selects the
    null_policy="skip" window algorithm (faer_rolling_skipping_lr) with min_valid_rows.
    """
    n_features = len(x) + int(bool(add_bias))
    if window_size < 2:
        raise ValueError("`window_size` must be >= 2.")  # expr_linear.py:524-526
    if n_features > window_size:
        raise ValueError("# features > window size. Linear regression is not well-defined.")  # :527-531
    min_size = 0
    if skip_non_finite:
        min_size = min(n_features, window_size) if min_valid_rows is None else int(min_valid_rows)  # :535-543
        if min_size < n_features or min_size > window_size:
            raise ValueError("`min_valid_rows` must be in [#features, window_size].")
    return _windowed("pds_rolling_lr", x, target, window_size, add_bias, l2_reg, min_size, ctx)


def recursive_lin_reg(*x, target, start_with: int, add_bias: bool = False, l2_reg: float = 0.0, ctx: Context | None = None,
                      seed_moments=None):
    """
    pds.recursive_lin_reg (pl_recursive_lr): expanding-window fit from row start_with-1 on.
    seed_moments (the gram_moments matrix of rows that precede this frame) continues an earlier frame: the
    row-sharded multi-GPU form, see parallel.recursive_lin_reg_row_sharded.
    """
    n_features = len(x) + int(bool(add_bias))
    if start_with < n_features:
        raise ValueError("# features > number of rows for the initial fit.")  # expr_linear.py:455-459
    return _windowed("pds_recursive_lr", x, target, start_with, add_bias, l2_reg, 0, ctx, seed_moments=seed_moments)


def gram_moments(*x, target, weights=None, ctx: Context | None = None, out_device: bool = False):
    """The augmented moment matrix A = Z'Z, Z = [x1..xp | 1 | y] ((p+2)^2), the measured Gram build."""
    ctx = ctx or default_context()
    cols = _Cols(target, x, weights)
    _follow(ctx, cols)
    q = cols.n_feat + 2
    if out_device:
        if cols.space != _lib.PDS_DEVICE:
            raise ValueError("out_device requires device-resident inputs")
        out, out_p = _out_like(cols, (q, q))
        _lib.check(ctx.fn("pds_moments")(ctx._h, cols.cols, cols.weights, cols.n_feat, C.c_int64(cols.n_rows), cols.space,
                                         out_p, _lib.PDS_DEVICE))
        return out.t()  # column-major -> logical (symmetric anyway)
    out = np.empty((q, q), dtype=_dtype())
    _lib.check(ctx.fn("pds_moments")(ctx._h, cols.cols, cols.weights, cols.n_feat, C.c_int64(cols.n_rows), cols.space,
                                     C.c_void_p(out.ctypes.data), _lib.PDS_HOST))
    return out.T


def lin_reg_from_moments(moments, *, add_bias: bool = False, l1_reg: float = 0.0, l2_reg: float = 0.0, tol: float = 1e-5,
                         solver: str = "qr", max_iter: int = 200, positive: bool = False,
                         singular_x_tol: float | None = None, ctx: Context | None = None):
    """Solve from an (all-reduced) moment matrix: the replicated tail of the multi-GPU single-OLS path."""
    ctx = ctx or default_context()
    prm = _params(add_bias, l1_reg, l2_reg, tol, solver, positive, max_iter, singular_x_tol)
    if _is_torch(moments):
        import torch

        tdt = torch.float64 if config.LIN_REG_EXPR_F64 else torch.float32
        m = moments.to(tdt).t().contiguous() if moments.is_cuda else None
        if m is None:
            moments = moments.numpy()
    if _is_torch(moments):
        q = int(m.shape[0])
        mp, space = C.c_void_p(int(m.data_ptr())), _lib.PDS_DEVICE
    else:
        m = np.asfortranarray(np.asarray(moments, dtype=_dtype()))
        q = int(m.shape[0])
        mp, space = C.c_void_p(m.ctypes.data), _lib.PDS_HOST
    p = q - 2
    pp = p + int(bool(add_bias))
    coeffs = np.empty(pp, dtype=_dtype())
    is_null = C.c_int(0)
    _lib.check(ctx.fn("pds_lr_from_moments")(ctx._h, mp, space, p, C.byref(prm), C.c_void_p(coeffs.ctypes.data),
                                             C.byref(is_null)))
    return None if is_null.value else coeffs
