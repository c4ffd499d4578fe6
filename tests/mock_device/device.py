"""
Python callbacks behind the mock trampolines: every `pds_*` entry point the plugin layer calls, answered by the CPU oracle
(oracle/oracle.py) with the argument meaning and error behaviour include/pds_lstsq.h documents.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import oracle as orc  # noqa: E402
from polars_ds_extension_amd._lib import LRParams  # noqa: E402  (ctypes mirror of pds_lr_params; no library is loaded)

from . import build as _build  # noqa: E402

OK, INVALID, EMPTY, TOO_FEW, UNSUPPORTED, NULLS = 0, -1, -2, -3, -5, -7
RAISE, SKIP, FILL, IGNORE = 0, 1, 2, 3
SOLVER = {0: "qr", 1: "svd", 2: "choleskey"}
SE = {0: "se", 1: "hc0", 2: "hc1", 3: "hc2", 4: "hc3"}
CT = {"f64": (C.c_double, np.float64), "f32": (C.c_float, np.float32)}
_C2CT = {"int": C.c_int, "int64_t": C.c_int64, "double": C.c_double, "float": C.c_float, "pds_space": C.c_int, "size_t": C.c_size_t}


class MockError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code, self.msg = code, msg


def _view(ptr, n, dt):
    if not ptr or n == 0:
        return np.empty(0, dtype=dt)
    return np.ctypeslib.as_array((np.ctypeslib.as_ctypes_type(dt) * int(n)).from_address(ptr))


def _columns(cols_ptr, nc, n, dt):
    table = (C.c_void_p * nc).from_address(cols_ptr)
    return [_view(table[i], n, dt) for i in range(nc)]


def _bit(bm_ptr, off, n):
    """validity bits [off, off + n) of an Arrow bitmap as a bool array (None pointer = all valid)."""
    if not bm_ptr:
        return np.ones(n, dtype=bool)
    raw = _view(bm_ptr, (off + n + 7) // 8, np.uint8)
    return np.unpackbits(raw, bitorder="little")[off:off + n].astype(bool)


def _validity(val_ptr, off_ptr, nc, n):
    if not val_ptr:
        return [np.ones(n, dtype=bool) for _ in range(nc)]
    vt = (C.c_void_p * nc).from_address(val_ptr)
    offs = _view(off_ptr, nc, np.int64) if off_ptr else np.zeros(nc, dtype=np.int64)
    return [_bit(vt[i], int(offs[i]), n) for i in range(nc)]


def _prm(ptr):
    p = LRParams.from_address(ptr)
    return dict(add_bias=bool(p.add_bias), l1_reg=p.l1_reg, l2_reg=p.l2_reg, tol=p.tol, solver=SOLVER.get(p.solver, "qr"),
                positive=bool(p.positive), max_iter=p.max_iter, singular_x_tol=p.singular_x_tol)


def _check_shape(n_feat, n, bias):
    if n_feat < 1:
        raise MockError(INVALID, "need at least one feature column")
    if n == 0:
        raise MockError(EMPTY, "Empty data")
    if n < n_feat + int(bias):
        raise MockError(TOO_FEW, "#Data < #features. No conclusive result.")


def _fit(X, y, prm, f32, weights=None):
    kw = dict(prm)
    if weights is not None:
        return orc.pl_lr(X, y, add_bias=kw["add_bias"], solver=kw["solver"], weights=weights)
    return orc.pl_lr(X, y, f32_path=f32, **kw)


def _apply_policy(cols, valid, policy, fill, dt):
    """series_to_mat_for_lr (linear_regression.rs:187-266) on [y, x...]: returns (columns, keep mask)."""
    n = len(cols[0])
    any_null = any((~v).any() for v in valid)
    if not any_null:
        return [np.asarray(c) for c in cols], np.ones(n, dtype=bool)
    if policy == RAISE:
        raise MockError(NULLS, "Nulls found in data")
    keep = np.ones(n, dtype=bool)
    out = [np.array(c, dtype=dt) for c in cols]
    if policy == SKIP:
        for v in valid:
            keep &= v
    elif policy == FILL:
        keep &= valid[0]
        for c, v in zip(out[1:], valid[1:]):
            c[~v] = fill
    elif policy == IGNORE:
        for c, v in zip(out, valid):
            c[~v] = np.nan
    else:
        raise MockError(INVALID, "Invalid NullPolicy.")
    return [c[keep] for c in out], keep


def make_callbacks(sfx):
    ct, dt = CT[sfx]
    f32 = sfx == "f32"

    def X_of(cols):
        return np.column_stack(cols[1:]) if len(cols) > 1 else np.empty((len(cols[0]), 0), dtype=dt)

    def put_fit(b, pp, coeffs_p, is_null_p):
        if is_null_p:
            C.c_int.from_address(is_null_p).value = int(b is None)
        _view(coeffs_p, pp, dt)[:] = np.nan if b is None else b
        return b

    def lr(ctx, cols_p, w_p, n_feat, n, space, prm_p, coeffs_p, is_null_p, pred_p=0, resid_p=0):
        prm = _prm(prm_p)
        _check_shape(n_feat, n, prm["add_bias"])
        cols = _columns(cols_p, n_feat + 1, n, dt)
        X, y = X_of(cols), cols[0]
        b = put_fit(_fit(X, y, prm, f32, _view(w_p, n, dt) if w_p else None), n_feat + prm["add_bias"], coeffs_p, is_null_p)
        if pred_p or resid_p:
            Xb = orc.with_bias(X) if prm["add_bias"] else X
            pr = (Xb @ b).astype(dt) if b is not None else np.full(n, np.nan, dtype=dt)
            if pred_p:
                _view(pred_p, n, dt)[:] = pr
            if resid_p:
                _view(resid_p, n, dt)[:] = y - pr
        return OK

    def elastic_net(ctx, cols_p, n_feat, n, space, prm_p, coeffs_p):
        prm = _prm(prm_p)
        if n == 0:
            raise MockError(EMPTY, "Empty data")
        cols = _columns(cols_p, n_feat + 1, n, dt)
        X = X_of(cols)
        Xb = orc.with_bias(X) if prm["add_bias"] else X
        b = orc.coordinate_descent(Xb, cols[0], max(prm["l1_reg"], 0.0), max(prm["l2_reg"], 0.0), bool(prm["add_bias"]), prm["tol"],
                                   prm["max_iter"], False)
        _view(coeffs_p, n_feat + prm["add_bias"], dt)[:] = b
        return OK

    def lr_pred(ctx, cols_p, w_p, n_feat, n, space, prm_p, coeffs_p, is_null_p, pred_p, resid_p):
        return lr(ctx, cols_p, w_p, n_feat, n, space, prm_p, coeffs_p, is_null_p, pred_p, resid_p)

    def lr_nullable(ctx, cols_p, val_p, off_p, n_feat, n, space, policy, fill, prm_p, coeffs_p, is_null_p, pred_p, resid_p,
                    row_valid_p, n_used_p):
        prm = _prm(prm_p)
        if n_feat < 1:
            raise MockError(INVALID, "need at least one feature column")
        if n == 0:
            raise MockError(EMPTY, "Empty data")
        if policy < RAISE or policy > IGNORE:
            raise MockError(INVALID, "Invalid NullPolicy.")
        cols = _columns(cols_p, n_feat + 1, n, dt)
        kept, keep = _apply_policy(cols, _validity(val_p, off_p, n_feat + 1, n), policy, fill, dt)
        nk = int(keep.sum())
        if n_used_p:
            C.c_int64.from_address(n_used_p).value = nk
        pp = n_feat + prm["add_bias"]
        if nk == 0:
            raise MockError(EMPTY, "Empty data")
        if nk < pp:
            raise MockError(TOO_FEW, "#Data < #features. No conclusive result.")
        X, y = X_of(kept), kept[0]
        b = put_fit(_fit(X, y, prm, f32), pp, coeffs_p, is_null_p)
        if pred_p or resid_p:
            Xb = orc.with_bias(X) if prm["add_bias"] else X
            pr = (Xb @ b).astype(dt) if b is not None else np.full(nk, np.nan, dtype=dt)
            full_p, full_r = np.full(n, np.nan, dtype=dt), np.full(n, np.nan, dtype=dt)
            full_p[keep], full_r[keep] = pr, y - pr
            if pred_p:
                _view(pred_p, n, dt)[:] = full_p
            if resid_p:
                _view(resid_p, n, dt)[:] = full_r
            if row_valid_p:
                _view(row_valid_p, n, np.uint8)[:] = keep
        return OK

    def multi(ctx, cols_p, k, n_feat, n, space, bias, l2, solver, tol, coeffs_p, is_null_p, pred_p, resid_p):
        if n == 0:
            raise MockError(EMPTY, "Empty data")
        pp = n_feat + int(bool(bias))
        if n < pp:
            raise MockError(TOO_FEW, "#Data < #features. No conclusive result.")
        cols = _columns(cols_p, k + n_feat, n, dt)
        X = np.column_stack(cols[k:])
        Xb = orc.with_bias(X) if bias else X
        co = _view(coeffs_p, k * pp, dt).reshape(k, pp)
        gated = False
        for t in range(k):
            b = orc.pl_lr(X, cols[t], add_bias=bool(bias), l2_reg=l2, solver=SOLVER.get(solver, "qr"), singular_x_tol=tol)
            gated |= b is None
            co[t] = np.nan if b is None else b
            if pred_p:
                pr = (Xb @ b).astype(dt) if b is not None else np.full(n, np.nan, dtype=dt)
                _view(pred_p, k * n, dt).reshape(k, n)[t] = pr
                _view(resid_p, k * n, dt).reshape(k, n)[t] = cols[t] - pr
        if is_null_p:
            C.c_int.from_address(is_null_p).value = int(gated)
        return OK

    def rcond(ctx, cols_p, n_feat, n, space, bias, l2, rc, coeffs_p, sv_p):
        _check_shape(n_feat, n, bias)
        cols = _columns(cols_p, n_feat + 1, n, dt)
        X = X_of(cols)
        b, s = orc.solve_lr_rcond(orc.with_bias(X) if bias else X, cols[0], l2, bool(bias), rc)
        pp = n_feat + int(bool(bias))
        _view(coeffs_p, pp, dt)[:] = b
        _view(sv_p, pp, dt)[:] = s
        return OK

    class Report(C.Structure):
        _fields_ = [(k, C.c_void_p) for k in ("beta", "std_err", "t", "p", "ci_lower", "ci_upper")] + [("r2", ct), ("adj_r2", ct)]

    def put_report(out_p, rep, pp):
        r = Report.from_address(out_p)
        for name, key in (("beta", "beta"), ("std_err", "std_err"), ("t", "t"), ("p", "p"), ("ci_lower", "ci_lo"), ("ci_upper", "ci_hi")):
            _view(getattr(r, name), pp, dt)[:] = rep[key]
        r.r2, r.adj_r2 = float(rep["r2"]), float(rep["adj_r2"])

    def run_report(X, y, w, bias, se_type, y_var, out_p):
        Xb = orc.with_bias(X) if bias else X
        yv = y_var  # (as given: a NaN propagates into r2 / adj_r2 like a null target.var() does in the reference)
        rep = orc.wls_report(Xb, y, w, y_var=yv) if w is not None else orc.lin_reg_report(Xb, y, y_var=yv, std_err=SE[se_type])
        put_report(out_p, rep, Xb.shape[1])

    def report(ctx, cols_p, w_p, n_feat, n, space, bias, se_type, y_var, out_p):
        _check_shape(n_feat, n, bias)
        cols = _columns(cols_p, n_feat + 1, n, dt)
        if (se_type & 0x100) and not w_p:  # PDS_REPORT_DERIVE_YVAR: the library takes target.var() (ddof = 1) from its own pass
            y_var = float(np.var(np.asarray(cols[0], dtype=np.float64), ddof=1))
        se_type &= ~0x100
        run_report(X_of(cols), cols[0], _view(w_p, n, dt) if w_p else None, bool(bias), se_type, y_var, out_p)
        return OK

    def report_nullable(ctx, cols_p, val_p, off_p, n_feat, n, space, policy, fill, bias, se_type, y_var, out_p, n_used_p):
        if n == 0:
            raise MockError(EMPTY, "Empty data")
        cols = _columns(cols_p, n_feat + 1, n, dt)
        kept, keep = _apply_policy(cols, _validity(val_p, off_p, n_feat + 1, n), policy, fill, dt)
        nk = int(keep.sum())
        if n_used_p:
            C.c_int64.from_address(n_used_p).value = nk
        _check_shape(n_feat, nk, bias)
        run_report(X_of(kept), kept[0], None, bool(bias), se_type, y_var, out_p)
        return OK

    def grouped_core(cols, off, prm, coeffs_p, null_p, weights=None, valid=None, policy=RAISE, fill=0.0):
        ng = len(off) - 1
        pp = len(cols) - 1 + prm["add_bias"]
        co = _view(coeffs_p, ng * pp, dt).reshape(ng, pp)
        nu = _view(null_p, ng, np.uint8)
        for g in range(ng):
            s = slice(int(off[g]), int(off[g + 1]))
            gc = [c[s] for c in cols]
            if valid is not None:
                gc, _ = _apply_policy(gc, [v[s] for v in valid], policy, fill, dt)
            b = None
            if len(gc[0]) >= pp and len(gc[0]) > 0:
                b = _fit(X_of(gc), gc[0], prm, f32, None if weights is None else weights[s])
            co[g] = np.nan if b is None else b
            nu[g] = b is None
        return OK

    def grouped(ctx, cols_p, n_feat, n, off_p, ng, space, prm_p, coeffs_p, null_p):
        if ng <= 0 or n <= 0:
            raise MockError(EMPTY, "Empty data")
        return grouped_core(_columns(cols_p, n_feat + 1, n, dt), _view(off_p, ng + 1, np.int64), _prm(prm_p), coeffs_p, null_p)

    def grouped_weighted(ctx, cols_p, w_p, n_feat, n, off_p, ng, space, prm_p, coeffs_p, null_p):
        if ng <= 0 or n <= 0:
            raise MockError(EMPTY, "Empty data")
        return grouped_core(_columns(cols_p, n_feat + 1, n, dt), _view(off_p, ng + 1, np.int64), _prm(prm_p), coeffs_p, null_p,
                            weights=_view(w_p, n, dt))

    def grouped_nullable(ctx, cols_p, val_p, off_bits_p, n_feat, n, off_p, ng, space, policy, fill, prm_p, coeffs_p, null_p):
        if ng <= 0 or n <= 0:
            raise MockError(EMPTY, "Empty data")
        valid = _validity(val_p, off_bits_p, n_feat + 1, n)
        if policy == RAISE and any((~v).any() for v in valid):
            raise MockError(NULLS, "Nulls found in data")
        return grouped_core(_columns(cols_p, n_feat + 1, n, dt), _view(off_p, ng + 1, np.int64), _prm(prm_p), coeffs_p, null_p,
                            valid=valid, policy=policy, fill=fill)

    def by_key(ctx, cols_p, keys_p, n_feat, n, space, prm_p, max_groups, out_keys_p, coeffs_p, null_p, ng_p):
        if n <= 0:
            raise MockError(EMPTY, "Empty data")
        if max_groups < 1:
            raise MockError(INVALID, "max_groups must be positive")
        keys = _view(keys_p, n, np.int64)
        order = np.argsort(keys, kind="stable")
        sk = keys[order]
        uniq, start = np.unique(sk, return_index=True)
        ng = len(uniq)
        C.c_int64.from_address(ng_p).value = ng
        if ng > max_groups:
            raise MockError(INVALID, "more distinct keys than max_groups")
        cols = [c[order] for c in _columns(cols_p, n_feat + 1, n, dt)]
        _view(out_keys_p, ng, np.int64)[:] = uniq
        return grouped_core(cols, np.concatenate([start, [n]]).astype(np.int64), _prm(prm_p), coeffs_p, null_p)

    def pred_rows(cols, off, bias, co, nu, order, pred_p, resid_p, rn_p):
        """pred / resid / row_null of the rows of `cols` (group order); `order` (sorted position -> frame row) or None"""
        n = len(cols[0])
        X = X_of(cols)
        Xb = orc.with_bias(X) if bias else X
        gid = np.repeat(np.arange(len(off) - 1), np.diff(off))
        with np.errstate(invalid="ignore"):
            pr = np.einsum("ij,ij->i", np.asarray(Xb, dtype=np.float64), np.asarray(co, dtype=np.float64)[gid]).astype(dt)
        rn = np.asarray(nu, dtype=bool)[gid]
        pr[rn] = np.nan
        re = (np.asarray(cols[0], dtype=np.float64) - pr.astype(np.float64)).astype(dt)
        dst = np.arange(n) if order is None else order
        if pred_p:
            _view(pred_p, n, dt)[dst] = pr
        if resid_p:
            _view(resid_p, n, dt)[dst] = re
        if rn_p:
            _view(rn_p, n, np.uint8)[dst] = rn

    def _own(ptr, count, dtype):
        """the caller's buffer, or scratch of the same shape when the caller passed NULL"""
        arr = np.empty(count, dtype=dtype)
        return (ptr, None) if ptr else (arr.ctypes.data, arr)

    def grouped_pred(ctx, cols_p, w_p, n_feat, n, off_p, ng, space, prm_p, coeffs_p, null_p, pred_p, resid_p, rn_p):
        if ng <= 0 or n <= 0:
            raise MockError(EMPTY, "Empty data")
        if not (pred_p or resid_p or rn_p):
            raise MockError(INVALID, "pds_lr_grouped_pred: no per-row output requested")
        prm = _prm(prm_p)
        pp = n_feat + prm["add_bias"]
        cols = _columns(cols_p, n_feat + 1, n, dt)
        off = _view(off_p, ng + 1, np.int64)
        cp, keep1 = _own(coeffs_p, ng * pp, dt)
        npn, keep2 = _own(null_p, ng, np.uint8)
        if w_p:
            prm = dict(prm, l1_reg=0.0, l2_reg=0.0, positive=0, singular_x_tol=0.0)
        grouped_core(cols, off, prm, cp, npn, weights=_view(w_p, n, dt) if w_p else None)
        pred_rows(cols, off, prm["add_bias"], _view(cp, ng * pp, dt).reshape(ng, pp), _view(npn, ng, np.uint8), None, pred_p, resid_p, rn_p)
        return OK

    def by_key_pred(ctx, cols_p, w_p, keys_p, n_feat, n, space, prm_p, max_groups, out_keys_p, coeffs_p, null_p, ng_p, pred_p, resid_p,
                    rn_p):
        if n <= 0:
            raise MockError(EMPTY, "Empty data")
        want_coef = bool(out_keys_p or coeffs_p)
        if not want_coef:
            max_groups = n
        if max_groups < 1:
            raise MockError(INVALID, "max_groups must be positive")
        prm = _prm(prm_p)
        pp = n_feat + prm["add_bias"]
        keys = _view(keys_p, n, np.int64)
        order = np.argsort(keys, kind="stable")
        uniq, start = np.unique(keys[order], return_index=True)
        ng = len(uniq)
        if ng_p:
            C.c_int64.from_address(ng_p).value = ng
        if ng > max_groups:
            raise MockError(INVALID, "more distinct keys than max_groups")
        cols = [c[order] for c in _columns(cols_p, n_feat + 1, n, dt)]
        off = np.concatenate([start, [n]]).astype(np.int64)
        if out_keys_p:
            _view(out_keys_p, ng, np.int64)[:] = uniq
        cp, keep1 = _own(coeffs_p, ng * pp, dt)
        npn, keep2 = _own(null_p, ng, np.uint8)
        if w_p:
            prm = dict(prm, l1_reg=0.0, l2_reg=0.0, positive=0, singular_x_tol=0.0)
        grouped_core(cols, off, prm, cp, npn, weights=_view(w_p, n, dt)[order] if w_p else None)
        if pred_p or resid_p or rn_p:
            pred_rows(cols, off, prm["add_bias"], _view(cp, ng * pp, dt).reshape(ng, pp), _view(npn, ng, np.uint8), order, pred_p, resid_p,
                      rn_p)
        return OK

    def by_key_multi(ctxs_p, n_ctx, n_slices, cols_p, keys_p, n_feat, n, prm_p, max_groups, out_keys_p, coeffs_p, null_p, ng_p):
        return by_key(None, cols_p, keys_p, n_feat, n, 0, prm_p, max_groups, out_keys_p, coeffs_p, null_p, ng_p)

    def by_key_pred_multi(ctxs_p, n_ctx, n_slices, cols_p, w_p, keys_p, n_feat, n, prm_p, pred_p, resid_p, rn_p):
        return by_key_pred(None, cols_p, w_p, keys_p, n_feat, n, 0, prm_p, n, None, None, None, None, pred_p, resid_p, rn_p)

    def windowed(cols, n_feat, n, bias, lam, coeffs_p, pred_p, valid_p, first_valid, rows):
        """rows: (coefficient rows for output rows first_valid.., validity of those rows)"""
        pp = n_feat + int(bool(bias))
        co = _view(coeffs_p, n * pp, dt).reshape(n, pp)
        pr, va = _view(pred_p, n, dt), _view(valid_p, n, np.uint8)
        co[:], pr[:], va[:] = np.nan, np.nan, 0
        b, ok = rows
        X = X_of(cols)
        Xb = orc.with_bias(X) if bias else X
        m = len(b)
        co[first_valid:first_valid + m] = b
        va[first_valid:first_valid + m] = ok
        with np.errstate(invalid="ignore"):
            pr[first_valid:first_valid + m] = np.einsum("ij,ij->i", Xb[first_valid:first_valid + m], b)
        bad = va == 0
        co[bad], pr[bad] = np.nan, np.nan

    def rolling(ctx, cols_p, n_feat, n, space, bias, window, min_size, lam, coeffs_p, pred_p, valid_p):
        _check_shape(n_feat, n, bias)
        cols = _columns(cols_p, n_feat + 1, n, dt)
        X = X_of(cols)
        Xb = orc.with_bias(X) if bias else X
        if window > n:
            windowed(cols, n_feat, n, bias, lam, coeffs_p, pred_p, valid_p, 0, (np.empty((0, Xb.shape[1]), dtype=dt), np.empty(0, dtype=bool)))
            return OK
        if min_size > 0:
            b, ok = orc.rolling_skipping_lr(Xb, cols[0], int(window), int(min_size), lam)
            full_b = np.full((n - window + 1, Xb.shape[1]), np.nan, dtype=dt)
            full_ok = np.zeros(n - window + 1, dtype=bool)
            full_b[len(full_b) - len(b):], full_ok[len(full_ok) - len(ok):] = b, ok  # the skipping driver starts late (:249-272)
            rows = (full_b, full_ok)
        else:
            b = orc.rolling_lr(Xb, cols[0], int(window), lam)
            rows = (b, np.ones(len(b), dtype=bool))
        windowed(cols, n_feat, n, bias, lam, coeffs_p, pred_p, valid_p, int(window) - 1, rows)
        return OK

    def recursive(ctx, cols_p, n_feat, n, space, bias, start_with, lam, coeffs_p, pred_p, valid_p):
        _check_shape(n_feat, n, bias)
        cols = _columns(cols_p, n_feat + 1, n, dt)
        X = X_of(cols)
        Xb = orc.with_bias(X) if bias else X
        if start_with > n:
            windowed(cols, n_feat, n, bias, lam, coeffs_p, pred_p, valid_p, 0, (np.empty((0, Xb.shape[1]), dtype=dt), np.empty(0, dtype=bool)))
            return OK
        b = orc.recursive_lr(Xb, cols[0], int(start_with), lam)
        windowed(cols, n_feat, n, bias, lam, coeffs_p, pred_p, valid_p, int(start_with) - 1, (b, np.ones(len(b), dtype=bool)))
        return OK

    def moments(ctx, cols_p, w_p, n_feat, n, space, out_p, out_space):
        _check_shape(n_feat, n, 0)
        cols = _columns(cols_p, n_feat + 1, n, dt)
        Z = np.column_stack(cols[1:] + [np.ones(n, dtype=dt), cols[0]]).astype(np.float64)
        A = (Z * _view(w_p, n, dt)[:, None]).T @ Z if w_p else Z.T @ Z
        q = n_feat + 2
        _view(out_p, q * q, dt)[:] = A.T.reshape(-1)  # column-major
        return OK

    def from_moments(ctx, mom_p, mom_space, n_feat, prm_p, coeffs_p, is_null_p):
        prm = _prm(prm_p)
        q, p, bias = n_feat + 2, n_feat, prm["add_bias"]
        pp = p + bias
        A = _view(mom_p, q * q, dt).reshape(q, q).T.astype(dt)
        G = np.ascontiguousarray(A[:pp, :pp]).copy()
        c = np.ascontiguousarray(A[:pp, p + 1]).copy()
        meth = orc.lr_methods(prm["l1_reg"], prm["l2_reg"])
        if meth in ("normal", "l2") and not prm["positive"]:
            G[np.arange(p), np.arange(p)] += prm["l2_reg"]
            b = orc.gated_solve_gram(G, c, prm["solver"], prm["singular_x_tol"]) if prm["singular_x_tol"] > 0 else orc.solve_gram(G, c, prm["solver"])
        elif meth == "normal":
            raise MockError(UNSUPPORTED, "mock: NNLS from moments is not modelled")
        else:
            b = orc.cd_from_gram(G, c, A[:p, p].copy(), float(A[p, p + 1]), float(A[p, p]), prm["l1_reg"], prm["l2_reg"], bias, prm["tol"],
                                 2000 if f32 else prm["max_iter"], prm["positive"])[0]
        put_fit(None if b is None else np.asarray(b).reshape(-1), pp, coeffs_p, is_null_p)
        return OK

    def with_inv(ctx, cols_p, n_feat, n, space, bias, lam, coeffs_p, inv_p):
        _check_shape(n_feat, n, bias)
        cols = _columns(cols_p, n_feat + 1, n, dt)
        X = X_of(cols)
        Xb = np.asfortranarray(orc.with_bias(X) if bias else X, dtype=dt)
        pp = Xb.shape[1]
        inv = np.zeros((pp, pp), dtype=dt, order="F")
        beta = np.zeros(pp, dtype=dt)
        y = np.ascontiguousarray(cols[0], dtype=dt)
        fn = getattr(orc.lib(), "orc_qr_lr_with_inv_" + sfx)
        fn(Xb.ctypes.data_as(C.c_void_p), C.c_int64(n), y.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int(pp), ct(lam), C.c_int(int(bool(bias))),
           inv.ctypes.data_as(C.c_void_p), beta.ctypes.data_as(C.c_void_p))
        _view(coeffs_p, pp, dt)[:] = beta
        _view(inv_p, pp * pp, dt)[:] = inv.reshape(-1, order="F")
        return OK

    def glm_irls(ctx, cols_p, n_feat, n, space, bias, link, variance, tol, max_iter, coeffs_p, n_iter_p):
        if n_feat < 1 or n_feat > 16:
            raise MockError(UNSUPPORTED if n_feat > 16 else INVALID, "GLM (IRLS): up to 16 feature columns")
        if n <= 0:
            raise MockError(EMPTY, "Empty data")
        cols = _columns(cols_p, n_feat + 1, n, dt)
        X = X_of(cols)
        Xb = np.asfortranarray(orc.with_bias(X) if bias else X, dtype=dt)
        pp = Xb.shape[1]
        beta = np.zeros(pp, dtype=dt)
        y = np.ascontiguousarray(cols[0], dtype=dt)
        fn = getattr(orc.lib(), "orc_glm_irls_" + sfx)
        fn.restype = C.c_int
        it = fn(Xb.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int(pp), C.c_int(link), C.c_int(variance),
                ct(tol), C.c_int(max_iter), beta.ctypes.data_as(C.c_void_p))
        _view(coeffs_p, pp, dt)[:] = beta
        if n_iter_p:
            _view(n_iter_p, 1, np.int32)[0] = it
        return OK

    return {f"pds_glm_irls_{sfx}": glm_irls, f"pds_lr_with_inv_{sfx}": with_inv, f"pds_moments_{sfx}": moments, f"pds_lr_from_moments_{sfx}": from_moments, f"pds_lr_{sfx}": lr, f"pds_lr_pred_{sfx}": lr_pred, f"pds_lr_nullable_{sfx}": lr_nullable, f"pds_lr_multi_{sfx}": multi,
            f"pds_lr_rcond_{sfx}": rcond, f"pds_elastic_net_{sfx}": elastic_net, f"pds_lin_reg_report_{sfx}": report, f"pds_lin_reg_report_nullable_{sfx}": report_nullable,
            f"pds_lr_grouped_{sfx}": grouped, f"pds_lr_grouped_weighted_{sfx}": grouped_weighted,
            f"pds_lr_grouped_nullable_{sfx}": grouped_nullable, f"pds_lr_by_key_{sfx}": by_key, f"pds_lr_grouped_pred_{sfx}": grouped_pred,
            f"pds_lr_by_key_pred_{sfx}": by_key_pred, f"pds_lr_by_key_multi_{sfx}": by_key_multi, f"pds_lr_by_key_pred_multi_{sfx}": by_key_pred_multi, f"pds_rolling_lr_{sfx}": rolling,
            f"pds_recursive_lr_{sfx}": recursive}


_KEEP = []  # CFUNCTYPE objects must outlive the library


def load():
    """Build (if stale) and load the mock plugin library with every entry point bound.  Returns the CDLL."""
    import os

    # (PDS_MOCK_LIB: a sanitizer build of the same sources, see tests/mock_device/sanitize.sh)
    lib = C.CDLL(os.environ.get("PDS_MOCK_LIB") or str(_build.build()))
    lib.mock_set_error.argtypes = [C.c_char_p]
    impl = {}
    impl.update(make_callbacks("f64"))
    impl.update(make_callbacks("f32"))
    for ret, name, args in _build.prototypes():
        if name not in impl:
            continue
        argtypes = [(_C2CT.get(t.replace("const ", "").strip()) if "*" not in t else C.c_void_p) for t, _ in args]
        assert all(a is not None for a in argtypes), (name, args)
        proto = C.CFUNCTYPE(C.c_int, *argtypes)
        fn = impl[name]

        def guarded(*a, _fn=fn, _name=name):
            try:
                return _fn(*a)
            except MockError as e:
                lib.mock_set_error(e.msg.encode())
                return e.code
            except Exception as e:  # a bug in the mock itself: surface it through the plugin's error channel
                lib.mock_set_error(f"mock {_name}: {type(e).__name__}: {e}".encode())
                return INVALID

        cb = proto(guarded)
        _KEEP.append(cb)
        getattr(lib, "mock_bind_" + name)(C.cast(cb, C.c_void_p))
    return lib


def load_for_package():
    """The mock library with the return / argument types polars_ds_extension_amd/_lib.load() sets on the product library
    (tests/test_oracle_reference_suite.py swaps it in to pin the oracle against the reference's own test-suite)."""
    lib = load()
    lib.pds_last_error.restype = C.c_char_p
    lib.pds_version.restype = C.c_char_p
    lib.pds_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.pds_ctx_destroy.argtypes = [C.c_void_p]
    lib.pds_ctx_destroy.restype = None
    return lib
