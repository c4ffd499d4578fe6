"""Mirror of /root/reference/python/polars_ds/config.py (the only global the lstsq path reads)."""

LIN_REG_EXPR_F64 = True
"""If True every linear-regression entry point computes in f64 (symbols *_f64); if False in f32."""


def _which_lin_reg(x: str) -> str:
    # config.py:15-16 of the reference: selects `pl_lr` vs `pl_lr_f32`
    return x if LIN_REG_EXPR_F64 else f"{x}_f32"
