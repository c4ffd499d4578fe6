"""
ctypes binding of libpds_lstsq_hip.so (the C ABI of include/pds_lstsq.h).

The library is built in-tree by `polars_ds_extension_amd._build.build()` (hipcc, gfx950).  There is NO
CPU fallback: if the shared object is missing or the device is not an MI355X-class gfx950 GPU every
entry point raises -- a silently different code path would void the parity claims.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "csrc" / "libpds_lstsq_hip.so"

PDS_HOST, PDS_DEVICE = 0, 1
SOLVERS = {"qr": 0, "svd": 1, "choleskey": 2}  # any other string -> qr (src/linear/lr/mod.rs:17-26)
SE_TYPES = {"se": 0, "hc0": 1, "hc1": 2, "hc2": 3, "hc3": 4}
PDS_REPORT_DERIVE_YVAR = 0x100  # include/pds_lstsq.h

EXPORTS = [
    "pds_last_error", "pds_version", "pds_ctx_create", "pds_ctx_destroy", "pds_ctx_set_stream",
    "pds_ctx_synchronize", "pds_ctx_num_cus", "pds_ctx_set_option", "pds_set_host_staging", "pds_rows_to_cols_f64", "pds_rows_to_cols_f32", "pds_glm_irls_f64", "pds_glm_irls_f32", "pds_lr_rowmajor_f64", "pds_lr_rowmajor_f32", "pds_ctx_workspace_spills", "pds_ctx_workspace_bytes", "pds_ctx_set_timing", "pds_ctx_get_timing", "pds_ctx_get_timing_samples",
    "pds_lr_f64", "pds_lr_f32", "pds_lr_pred_f64", "pds_lr_pred_f32", "pds_lr_rcond_f64", "pds_lr_rcond_f32", "pds_elastic_net_f64", "pds_elastic_net_f32", "pds_lr_nullable_f64", "pds_lr_nullable_f32", "pds_lr_multi_f64", "pds_lr_multi_f32",
    "pds_lin_reg_report_f64", "pds_lin_reg_report_f32",
    "pds_report_fit_from_moments_f64", "pds_report_fit_from_moments_f32", "pds_report_partials_f64", "pds_report_partials_f32",
    "pds_report_finish_f64", "pds_report_finish_f32",
    "pds_lr_grouped_f64", "pds_lr_grouped_f32",
    "pds_rolling_lr_f64", "pds_rolling_lr_f32", "pds_recursive_lr_f64", "pds_recursive_lr_f32",
    "pds_recursive_lr_seeded_f64", "pds_recursive_lr_seeded_f32",
    "pds_lin_reg_report_nullable_f64", "pds_lin_reg_report_nullable_f32",
    "pds_lr_grouped_nullable_f64", "pds_lr_grouped_nullable_f32",
    "pds_lr_by_key_f64", "pds_lr_by_key_f32", "pds_lr_by_key_multi_f64", "pds_lr_by_key_multi_f32", "pds_lr_by_key_pred_multi_f64", "pds_lr_by_key_pred_multi_f32", "pds_host_alloc", "pds_host_free", "pds_device_count", "pds_device_alloc", "pds_device_free", "pds_ipc_export", "pds_ipc_open", "pds_ipc_close", "pds_signal_post", "pds_signal_wait", "pds_signal_wait_status", "pds_lr_grouped_pred_f64", "pds_lr_grouped_pred_f32", "pds_lr_by_key_pred_f64", "pds_lr_by_key_pred_f32", "pds_lr_grouped_weighted_f64", "pds_lr_grouped_weighted_f32",
    "pds_lr_with_inv_f64", "pds_lr_with_inv_f32",
    "pds_moments_f64", "pds_moments_f32", "pds_lr_from_moments_f64", "pds_lr_from_moments_f32",
    "pds_student_t_sf", "pds_student_t_ppf",
    "pds_allreduce_sum_f64", "pds_allreduce_sum_f32", "pds_scatter_rows_f64", "pds_scatter_rows_f32", "pds_gather_f64", "pds_gather_f32",
]


class LRParams(C.Structure):
    """pds_lr_params == the numeric fields of LRKwargs (src/num_ext/linear_regression.rs:27-45)."""

    _fields_ = [
        ("add_bias", C.c_int),
        ("l1_reg", C.c_double),
        ("l2_reg", C.c_double),
        ("tol", C.c_double),
        ("solver", C.c_int),
        ("positive", C.c_int),
        ("max_iter", C.c_int),
        ("singular_x_tol", C.c_double),
    ]


class ReportF64(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("beta", "std_err", "t", "p", "ci_lower", "ci_upper")] + [
        ("r2", C.c_double),
        ("adj_r2", C.c_double),
    ]


class ReportF32(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("beta", "std_err", "t", "p", "ci_lower", "ci_upper")] + [
        ("r2", C.c_float),
        ("adj_r2", C.c_float),
    ]


class PdsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[pds {code}] {msg}")
        self.code = code
        self.msg = msg


_lib = None


def load() -> C.CDLL:
    """dlopen the HIP library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.  If this library
        # were loaded first it would bind /opt/rocm's copy, torch would then bring a second runtime and
        # whichever initialises HSA second sees "no ROCm-capable device".  Importing torch first makes the
        # loader resolve our DT_NEEDED libamdhip64.so.7 to the copy that is already mapped.
        try:
            import torch  # noqa: F401
        except Exception:  # torch-less host (e.g. a Rust plugin process): the system runtime is the only one
            pass
        lib = C.CDLL(str(LIB_PATH))
        lib.pds_last_error.restype = C.c_char_p
        lib.pds_version.restype = C.c_char_p
        lib.pds_student_t_sf.restype = C.c_double
        lib.pds_student_t_sf.argtypes = [C.c_double, C.c_double]
        lib.pds_student_t_ppf.restype = C.c_double
        lib.pds_student_t_ppf.argtypes = [C.c_double, C.c_double]
        lib.pds_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.pds_ctx_destroy.argtypes = [C.c_void_p]
        lib.pds_ctx_destroy.restype = None
        lib.pds_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        lib.pds_ctx_synchronize.argtypes = [C.c_void_p]
        lib.pds_ctx_num_cus.argtypes = [C.c_void_p]
        lib.pds_ctx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_longlong]
        if hasattr(lib, "pds_set_host_staging"):
            lib.pds_set_host_staging.argtypes = [C.c_double, C.c_double]
        if hasattr(lib, "pds_ctx_workspace_spills"):  # (absent from older builds used in A/B runs)
            lib.pds_ctx_workspace_spills.argtypes = [C.c_void_p]
            lib.pds_ctx_workspace_spills.restype = C.c_longlong
        if hasattr(lib, "pds_ctx_workspace_bytes"):
            lib.pds_ctx_workspace_bytes.argtypes = [C.c_void_p, C.c_int]
            lib.pds_ctx_workspace_bytes.restype = C.c_longlong
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise PdsError(rc, load().pds_last_error().decode("utf-8", "replace"))


def missing_exports() -> list[str]:
    lib = load()
    return [s for s in EXPORTS if not hasattr(lib, s)]
