"""
Polars expression builders that route through libpds_lstsq_hip.so's `_polars_plugin_*` symbols.

Same call signatures as /root/reference/python/polars_ds/exprs/expr_linear.py (`lin_reg` :105-274 incl. the multi-target
form, `lin_reg_w_rcond` :356-410, `lin_reg_report` :561-631, `rolling_lin_reg` :482-558, `recursive_lin_reg` :413-479) plus the key-aware
`lin_reg(..., by=key)` of SURVEY.md 8(b) and `lin_reg_by_group`, the frame-level replacement of `group_by().agg(lin_reg)`.
Importing this module needs `polars` (>= 1.4), which is NOT installable in the build image: tests/test_polars_exprs.py runs
every builder end to end with tests/mini_polars standing in for the engine (it implements the documented plugin calling
convention and the per-group evaluation of `group_by().agg()`); against a REAL Polars the module is still unverified.
"""
from __future__ import annotations

from pathlib import Path
from typing import Any, List

from . import config as cfg

PLUGIN_PATH = Path(__file__).resolve().parent / "csrc" / "libpds_lstsq_hip.so"


def _pl():
    import polars as pl  # deferred: keeps the rest of the package importable without polars

    return pl


def _formula(s: Any):
    pl = _pl()
    if isinstance(s, str):
        return pl.sql_expr(s).alias(s)
    if isinstance(s, pl.Series):
        return pl.lit(s)
    if isinstance(s, pl.Expr):
        return s
    if hasattr(s, "__array__"):
        return pl.lit(pl.Series(values=s.__array__()))
    raise ValueError("Input can only be str or polars expression. The str must be valid SQL strings that polars can understand.")


def _plugin(symbol: str, args, kwargs, **flags):
    from polars.plugins import register_plugin_function

    return register_plugin_function(plugin_path=PLUGIN_PATH, function_name=cfg._which_lin_reg(symbol), args=args, kwargs=kwargs,
                                    pass_name_to_apply=True, **flags)


def _dtype():
    pl = _pl()
    return pl.Float64 if cfg.LIN_REG_EXPR_F64 else pl.Float32


def lin_reg(*x, target, add_bias: bool = False, weights=None, return_pred: bool = False, l1_reg: float = 0.0,
            l2_reg: float = 0.0, tol: float = 1e-5, solver: str = "qr", max_iter: int = 200, null_policy: str = "skip",
            positive: bool = False, singular_x_tol: float | None = None, by=None):
    if singular_x_tol is None:
        singular_x_tol = 1e-12 if cfg.LIN_REG_EXPR_F64 else 1e-6  # expr_linear.py:179-186
    if isinstance(target, list):  # expr_linear.py:188-233
        n_targets = len(target)
        if n_targets == 0:
            raise ValueError("If `target` is a list, it cannot be empty.")
        if n_targets == 1:
            return lin_reg(*x, target=target[0], add_bias=add_bias, weights=weights, return_pred=return_pred, l1_reg=l1_reg,
                           l2_reg=l2_reg, tol=tol, solver=solver, null_policy=null_policy, singular_x_tol=singular_x_tol)
        dt = _dtype()
        cols = [_formula(t).alias(f"target_{i}").cast(dt) for i, t in enumerate(target)]
        kwargs = {"bias": add_bias, "null_policy": null_policy, "solver": solver, "last_target_idx": n_targets, "l2_reg": l2_reg,
                  "singular_x_tol": singular_x_tol}
        cols.extend(_formula(z) for z in x)
        if return_pred:
            return _plugin("pl_lr_multi_pred", cols, kwargs).alias("lr_pred")
        return _plugin("pl_lr_multi", cols, kwargs, returns_scalar=True).alias("coeffs")
    if max_iter <= 0:
        raise ValueError("Input `max_iter` must be a positive.")
    weighted = weights is not None
    kwargs = {"bias": add_bias, "null_policy": null_policy, "l1_reg": l1_reg, "l2_reg": l2_reg, "solver": solver, "tol": tol,
              "max_iter": max_iter, "weighted": weighted, "positive": positive, "singular_x_tol": singular_x_tol}
    dt = _dtype()
    cols = ([_formula(weights).cast(dt).rechunk()] if weighted else []) + [_formula(target).cast(dt)] + [_formula(z) for z in x]
    if by is not None:  # one call computes every group (integer key column, any row order, nulls = one group)
        if return_pred:
            # Struct{pred, resid}, one row per input row in the frame's own order: what `.over(by)` / `group_by(by).agg(...)`
            # of `lin_reg(..., return_pred=True)` give per group (tests/test_linear_exprs.py:435-474)
            return _plugin("pl_lr_by_pred", [_formula(by), *cols], kwargs).alias("lr_pred")
        # Struct{key, coeffs}, one row per group, keys ascending
        return _plugin("pl_lr_by", [_formula(by), *cols], kwargs, changes_length=True).alias("coeffs_by")
    if return_pred:
        return _plugin("pl_lr_pred", cols, kwargs).alias("lr_pred")
    return _plugin("pl_lr", cols, kwargs, returns_scalar=True).alias("coeffs")


_GID = "__pds_gid"


def _is_integer_key(df, by) -> bool:
    if not isinstance(by, str):
        return False
    try:
        return bool(df.schema[by].is_integer())
    except Exception:
        return False


def _join(left, right, on, how="left"):
    """null keys must meet null keys (Polars' group_by makes them one group): `nulls_equal` (polars >= 1.24), `join_nulls` before"""
    try:
        return left.join(right, on=on, how=how, nulls_equal=True)
    except TypeError:
        return left.join(right, on=on, how=how, join_nulls=True)


def _with_group_ids(df, by):
    """
    Keys of any dtype (strings, dates, several columns, nulls) -> one dense integer id per row: the distinct key rows in order of
    first appearance get ids 0, 1, ... (`unique(maintain_order=True).with_row_index`), joined back onto the frame.  Returns
    (frame with the id column `__pds_gid`, the key table [id, *by]).  A single integer key column does not need this.
    """
    cols = [by] if isinstance(by, str) else list(by)
    # (the plugin returns its key field as Int64 and `with_row_index` makes UInt32: Polars refuses to join the two)
    keys = df.select(cols).unique(maintain_order=True).with_row_index(_GID).with_columns(_pl().col(_GID).cast(_pl().Int64))
    return _join(df, keys, on=cols), keys


def lin_reg_by_group(df, by, *x, target, return_pred: bool = False, **kwargs):
    """
    The replacement for `df.group_by(by).agg(pds.lin_reg(*x, target=...))` on this backend: ONE plugin call over the whole
    frame (`pl_lr_by`: keys in any row order, one fused kernel for every group) instead of one `pl_lr` call per group.
    An expression cannot know that it sits inside a `group_by` -- Polars hands the plugin one group's rows at a time, which
    costs a host-to-device round trip per group and runs BELOW the CPU reference (the coalescing queue of csrc/plugin.cpp
    only softens that) -- so the rewrite is explicit.
    `by`: one key column of any dtype or a list of key columns; null keys form one group, as in Polars.
    Returns a frame with one row per distinct key: columns *by and `coeffs` (a null list where the reference's per-group call
    returns a null list); integer keys come back ascending, other keys in order of first appearance.
    `return_pred=True` (tests/test_linear_exprs.py:435-474): the input frame with `pred` and `resid` columns, row for row.
    """
    if return_pred:
        if _is_integer_key(df, by):
            return df.with_columns(lin_reg(*x, target=target, by=by, return_pred=True, **kwargs)).unnest("lr_pred")
        ids, _ = _with_group_ids(df, by)
        return ids.with_columns(lin_reg(*x, target=target, by=_GID, return_pred=True, **kwargs)).unnest("lr_pred").drop(_GID)
    if _is_integer_key(df, by):
        # (the key field comes back Int64 under the key column's name: back to the frame's own integer dtype, so that callers
        #  -- lin_reg_over below -- can join it onto the frame)
        res = df.select(lin_reg(*x, target=target, by=by, **kwargs)).unnest("coeffs_by")
        return res.with_columns(_pl().col(by).cast(df.schema[by]))
    ids, keys = _with_group_ids(df, by)
    res = ids.select(lin_reg(*x, target=target, by=_GID, **kwargs)).unnest("coeffs_by")
    return _join(keys, res, on=[_GID]).drop(_GID)


def lin_reg_over(df, by, *x, target, **kwargs):
    """
    `df.with_columns(pds.lin_reg(*x, target=...).over(by))` (examples/basics.ipynb cells 16 / 18): every row carries its group's
    coefficient list.  One `pl_lr_by` call, then the per-group lists are joined back onto the frame.
    """
    cols = [by] if isinstance(by, str) else list(by)
    return _join(df, lin_reg_by_group(df, by, *x, target=target, **kwargs), on=cols)


def lin_reg_w_rcond(*x, target, add_bias: bool = False, rcond: float = 0.0, l2_reg: float = 0.0, null_policy: str = "raise"):
    """expr_linear.py:356-410: SVD solve with a singular-value cut-off; Struct{coeffs, singular_values}."""
    cols = [_formula(target).cast(_dtype())] + [_formula(z) for z in x]
    kwargs = {"bias": add_bias, "null_policy": null_policy, "l1_reg": 0.0, "l2_reg": l2_reg, "solver": "", "tol": abs(rcond)}
    return _plugin("pl_lr_w_rcond", cols, kwargs)


def lin_reg_report(*x, target, add_bias: bool = False, weights=None, std_err: str = "se", null_policy: str = "raise"):
    dt = _dtype()
    t = _formula(target).cast(dt)
    kwargs = {"bias": add_bias, "null_policy": null_policy, "std_err": std_err, "solver": "qr", "l1_reg": 0.0, "l2_reg": 0.0, "tol": 0.0}
    feats: List[Any] = [_formula(z) for z in x]
    if weights is None:
        return _plugin("pl_lin_reg_report", [t.var(), t, *feats], kwargs, changes_length=True).alias("lin_reg_report")
    return _plugin("pl_wls_report", [_formula(weights).cast(dt).rechunk(), t.var(), t, *feats], kwargs,
                   changes_length=True).alias("lin_reg_report")


def rolling_lin_reg(*x, target, window_size: int, add_bias: bool = False, l2_reg: float = 0.0, min_valid_rows: int | None = None,
                    null_policy: str = "raise"):
    n_features = len(x) + int(add_bias)
    if window_size < 2:
        raise ValueError("`window_size` must be >= 2.")
    if n_features > window_size:
        raise ValueError("# features > window size. Linear regression is not well-defined.")
    min_size = min(n_features, window_size) if min_valid_rows is None else int(min_valid_rows)
    kwargs = {"null_policy": null_policy, "n": window_size, "bias": add_bias, "lambda": abs(l2_reg), "min_size": min_size}
    cols = [_formula(target).cast(_dtype())] + [_formula(z) for z in x]
    return _plugin("pl_rolling_lr", cols, kwargs).alias("rolling_lin_reg")


def recursive_lin_reg(*x, target, start_with: int, add_bias: bool = False, l2_reg: float = 0.0, null_policy: str = "raise"):
    n_features = len(x) + int(add_bias)
    if start_with < n_features:
        raise ValueError("# features > number of rows for the initial fit.")
    kwargs = {"null_policy": null_policy, "n": start_with, "bias": add_bias, "lambda": abs(l2_reg), "min_size": 0}
    cols = [_formula(target).cast(_dtype())] + [_formula(z) for z in x]
    return _plugin("pl_recursive_lr", cols, kwargs).alias("recursive_lin_reg")


def query_ar_coeffs(x, lag: int, add_bias: bool = True, null_policy: str = "raise"):
    """exprs/ts_features.py:419-461: AR(lag) coefficients = lin_reg on shifted slices of the one series, bias last."""
    if null_policy not in ("raise", "one", "zero"):
        import math

        try:
            if not math.isfinite(float(null_policy)):
                raise ValueError
        except (TypeError, ValueError):
            raise ValueError("`null_polocy` must be 'raise', 'one', 'zero' or any finite numeric string for AR coefficients.") from None
    if lag <= 0:
        raise ValueError("`lag` must be > 0.")
    pl = _pl()
    xx = pl.col(x) if isinstance(x, str) else x
    return lin_reg(*[xx.shift(i).slice(offset=lag).alias(str(i)) for i in range(1, lag + 1)], target=xx.slice(offset=lag),
                   add_bias=add_bias, null_policy=null_policy)
