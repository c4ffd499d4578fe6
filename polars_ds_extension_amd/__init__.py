"""
polars_ds_extension_amd -- MI355X (gfx950) implementation of polars_ds's least-squares expression path.

Only that path (pds.lin_reg / lin_reg_report / rolling_lin_reg / recursive_lin_reg, OLS / ridge / lasso /
elastic net / NNLS / WLS) is implemented; see DESIGN.md for the scope table and INTEGRATION.md for how
the C ABI (include/pds_lstsq.h) drops in behind the reference's `#[polars_expr]` functions.
"""
from . import config  # noqa: F401
from . import linear_models  # noqa: F401  (LR / ElasticNet / OnlineLR over NumPy or CUDA tensors)
from .lstsq import (  # noqa: F401
    Context,
    DeviceBlock,
    GroupedFit,
    allreduce_sum,
    default_context,
    gather,
    gram_moments,
    lin_reg,
    lin_reg_by,
    lin_reg_by_key,
    lin_reg_by_key_multi,
    lin_reg_by_key_pred_multi,
    lin_reg_by_key_pred,
    lin_reg_by_pred,
    lin_reg_from_moments,
    lin_reg_report,
    lin_reg_w_rcond,
    query_ar_coeffs,
    recursive_lin_reg,
    rolling_lin_reg,
    scatter_rows,
)

__version__ = "0.1.0"
