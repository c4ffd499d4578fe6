/*
 * include/pds_lstsq.h -- C ABI of libpds_lstsq_hip.so, the MI355X (gfx950) implementation of the
 * polars_ds least-squares expression path.
 *
 * This is the drop-in boundary: the entry points are what the reference's Rust plugin functions
 * (`#[polars_expr] fn pl_lr / pl_lr_pred / pl_lin_reg_report / pl_rolling_lr / pl_recursive_lr`,
 * /root/reference/src/num_ext/linear_regression.rs:419,704,822,1121,1206 and the *_f32 twins in
 * linear_regression_f32.rs) would bind over `extern "C"` after unwrapping the Arrow column buffers.
 * They replace the calls those functions make into src/linear/lr/lr_solvers.rs and
 * src/linear/online_lr/lr_online_solvers.rs.  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in any signature.
 *   - `cols` is an array (in HOST memory) of n_feat + 1 column pointers in the reference's input
 *     order [y, x1, ..., xp] (expr_linear.py:250-258; weighted variants take `weights` separately).
 *     Each column is a contiguous buffer of n_rows values, exactly an Arrow Float64/Float32 values
 *     buffer.  `space` says where the column *buffers* live (PDS_DEVICE: already resident in HBM; PDS_HOST: host
 *     memory, copied to HBM by the library -- frames of more than one chunk (256 MiB, pds_set_host_staging) cross
 *     PCIe one row range at a time into a chunk-sized staging buffer for pds_lr_* / pds_moments_* (p <= 16), so HBM use
 *     is O(chunk) and the frame may be larger than HBM; the other entry points stage the whole frame).
 *   - output buffers are host memory unless the parameter is documented "space-resident".
 *   - every function returns PDS_OK (0) or a negative pds_status; pds_last_error() returns the
 *     message (thread-local), with the reference's error strings where the reference has one.
 *   - all launches go to the stream of the context (pds_ctx_set_stream); calls are synchronous with
 *     respect to the host on return (results are ready), except the *_async moment builders.
 *   - f64 symbols end in _f64, the f32 twins in _f32 (LIN_REG_EXPR_F64=False path, config.py:15-16).
 */
#ifndef PDS_LSTSQ_H
#define PDS_LSTSQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pds_ctx pds_ctx; /* opaque: device, stream, HBM workspace, pinned staging */

typedef enum { PDS_HOST = 0, PDS_DEVICE = 1 } pds_space;

typedef enum {
    PDS_OK = 0,
    PDS_ERR_INVALID = -1,     /* bad argument */
    PDS_ERR_EMPTY = -2,       /* "Empty data"                       linear_regression.rs:167 */
    PDS_ERR_TOO_FEW_ROWS = -3, /* "#Data < #features. No conclusive result."          :170-172 */
    PDS_ERR_HIP = -4,         /* HIP runtime failure / no gfx950 device */
    PDS_ERR_UNSUPPORTED = -5,
    PDS_ERR_NUMERIC = -6,     /* e.g. "SVD failed."                 lr_solvers.rs:256 */
    PDS_ERR_NULLS = -7        /* "Nulls found in data"              linear_regression.rs:198 */
} pds_status;

/* NullPolicy (src/linear/mod.rs:34-66) as the plugin functions use it for pl_lr / pl_lr_pred:
 * "raise" | "skip" | "zero"/"one"/numeric string -> FILL(value) | "ignore". */
typedef enum { PDS_NULL_RAISE = 0, PDS_NULL_SKIP = 1, PDS_NULL_FILL = 2, PDS_NULL_IGNORE = 3 } pds_null_policy;

/* LRSolverMethods::from(&str): "qr" (default, also any unknown string), "svd", "choleskey"
 * (src/linear/lr/mod.rs:17-26 -- the misspelling is the reference's). */
typedef enum { PDS_SOLVER_QR = 0, PDS_SOLVER_SVD = 1, PDS_SOLVER_CHOLESKEY = 2 } pds_solver;

/* StandardError::from(String): linear_regression.rs:113-132 */
typedef enum { PDS_SE = 0, PDS_HC0 = 1, PDS_HC1 = 2, PDS_HC2 = 3, PDS_HC3 = 4 } pds_se_type;
/* OR-ed into the se_type argument of pds_lin_reg_report_* (unweighted form): ignore `y_var` and take the target's sample
 * variance (ddof = 1) from the call's own pass over y (sums kept in f64 for both precisions).  Without the flag `y_var` is
 * used as given -- a NaN (what a null `target.var()` becomes, linear_regression.rs:836) propagates into r2 / adj_r2. */
#define PDS_REPORT_DERIVE_YVAR 0x100

/* ---- library / context ------------------------------------------------------------------- */
const char* pds_last_error(void);
const char* pds_version(void);
/* Creates a context on HIP device `device` (must be gfx950).  */
int pds_ctx_create(int device, pds_ctx** out);
void pds_ctx_destroy(pds_ctx* ctx);
/* Use an existing hipStream_t (passed as void*) for all work of this context instead of the private
 * non-blocking stream created with the context.  NULL selects HIP's default (null) stream. */
int pds_ctx_set_stream(pds_ctx* ctx, void* hip_stream);
int pds_ctx_synchronize(pds_ctx* ctx);
/* Number of compute units of the context's device (256 on MI355X). */
int pds_ctx_num_cus(const pds_ctx* ctx);
/* Behaviour switches of a context; the defaults are read from the environment ONCE, when the context is created:
 *   "keyed_sort"      (PDS_KEYED_SORT=1)      pds_lr_by_key_*: unordered keys always take the sorting route -- the determinism switch: the
 *                                             partition route appends records through cursor atomics, so its sums repeat to rounding only;
 *   "wide_f32_native" (PDS_WIDE_F32_NATIVE=1) f32 Gram builds beyond 64 features on the f32 matrix instructions (v_mfma_f32_32x32x2_f32,
 *                                             2.5e-7 from the f64 Gram) instead of three exact bf16 planes on the bf16 matrix cores (2e-6).
 * value: 0 / 1.  Unknown names are PDS_ERR_INVALID. */
int pds_ctx_set_option(pds_ctx* ctx, const char* name, long long value);
/* Host-frame staging (process wide): chunk_mb = bytes of one row chunk of a PDS_HOST frame (default 256, env
 * PDS_HOST_CHUNK_MB); resident_max_mb = largest host frame that pds_lr_pred_* still stages whole (one PCIe trip; larger
 * frames make two chunked trips; default 98304, env PDS_HOST_RESIDENT_MAX_MB).  A value <= 0 leaves the setting as is.
 * Replaces the reference's one-Vec marshalling copy, src/utils/mod.rs:101-206. */
int pds_set_host_staging(double chunk_mb, double resident_max_mb);

/* Diagnostics: how many call-local workspace slices had to be allocated outside the per-call reservation since the context
 * was created (0 unless an entry point under-estimated its bound; the slices are still valid, never out of bounds). */
long long pds_ctx_workspace_spills(const pds_ctx* ctx);
/* Bytes the context currently holds in one of its grow-only HBM workspaces: which = 0 call-local scratch, 1 host-frame staging,
 * 2 solver / key-order scratch, 3 the keyed-grouping workspace (pds_lr_by_key_*), 4 the weighted grouped frame.  -1 on a bad argument.
 * Lets a host size its frames against what a route reserves (the partition route of unordered keys must not reserve the sort
 * route's buffers: tests/test_gpu_parity.py). */
long long pds_ctx_workspace_bytes(const pds_ctx* ctx, int which);
/*
 * Measurement hooks (bench.py): with timing enabled every kernel class is bracketed by a HIP event
 * pair recorded on the context's stream.  pds_ctx_get_timing synchronises and returns, per class,
 * the summed duration in ms and the number of bracketed launches.  Classes: 0 Gram/moments (single
 * system, incl. its finalize), 1 grouped Gram, 2 batched solve, 3 residual pass, 4 rolling, 5 CD/NNLS.
 */
int pds_ctx_set_timing(pds_ctx* ctx, int enable);
int pds_ctx_get_timing(pds_ctx* ctx, double* ms_sum, long long* counts, int n_kinds, int reset);
/* The individual bracketed durations (ms) of one class, oldest first, at most `cap` (the newest are kept, up to 4096 per
 * class): returns how many were written, -1 on a bad argument.  bench.py reports min / median / max of the timed launches. */
int pds_ctx_get_timing_samples(pds_ctx* ctx, int kind, double* ms_out, int cap, int reset);

/* ---- LRKwargs mirror (linear_regression.rs:27-45), minus the strings parsed by the host ---- */
typedef struct {
    int add_bias;          /* kwargs.bias */
    double l1_reg;         /* kwargs.l1_reg */
    double l2_reg;         /* kwargs.l2_reg */
    double tol;            /* kwargs.tol (CD / NNLS convergence; rcond for pds_lr_rcond) */
    int solver;            /* pds_solver, from kwargs.solver.  Grouped fits with PDS_SOLVER_SVD and the rank gate on: the systems a
                            * chunk marks next to the gate go through a host SVD (the reference's thin_svd gate, lr_solvers.rs:358-366)
                            * while the marked set costs no more than 2048 systems of 66 x 66 (5.6e5 systems at 8 features); a larger
                            * set takes the device pivoted QR -- same gate statistic, beta from QR instead of the SVD. */
    int positive;          /* kwargs.positive */
    int max_iter;          /* kwargs.max_iter (the f32 twin ignores it: 2000 CD; NNLS 200 in pl_lr_f32, 2000 in pl_lr_pred_f32) */
    double singular_x_tol; /* kwargs.singular_x_tol: > 0 enables the log-det rank gate */
} pds_lr_params;

/*
 * pds_lr_*: the compute part of `pl_lr` (linear_regression.rs:419-513; f32: linear_regression_f32.rs
 * :289-384) on null-free columns: dispatch on (LRMethods::from((l1,l2)), positive) exactly as
 * :447-497, i.e. faer_solve_lr_gated / faer_solve_lr / faer_nn_lr / faer_coordinate_descent, or
 * faer_weighted_lr when `weights` != NULL (lr_solvers.rs:386-409).
 *   coeffs   out, n_feat + add_bias values (bias last).
 *   is_null  out, 1 when the rank gate fired (the reference returns a 1-row null list,
 *            linear_regression.rs:462-468); coeffs is then filled with NaN.
 */
int pds_lr_f64(pds_ctx* ctx, const double* const* cols, const double* weights, int n_feat,
               int64_t n_rows, pds_space space, const pds_lr_params* prm, double* coeffs,
               int* is_null);
int pds_lr_f32(pds_ctx* ctx, const float* const* cols, const float* weights, int n_feat,
               int64_t n_rows, pds_space space, const pds_lr_params* prm, float* coeffs,
               int* is_null);

/*
 * pds_lr_pred_*: `pl_lr_pred` (linear_regression.rs:704-820): same fit, then pred = X b and
 * resid = y - pred for every row.  pred/resid are `space`-resident buffers of n_rows values.
 */
int pds_lr_pred_f64(pds_ctx* ctx, const double* const* cols, const double* weights, int n_feat,
                    int64_t n_rows, pds_space space, const pds_lr_params* prm, double* coeffs,
                    int* is_null, double* pred, double* resid);
int pds_lr_pred_f32(pds_ctx* ctx, const float* const* cols, const float* weights, int n_feat,
                    int64_t n_rows, pds_space space, const pds_lr_params* prm, float* coeffs,
                    int* is_null, float* pred, float* resid);

/*
 * pds_elastic_net_*: ElasticNet::fit_unchecked of the model-class route (src/linear/lr/lr_solvers.rs:139-164, reached from
 * PyElasticNet.fit src/pymodels/py_lr.rs:112): ALWAYS faer_coordinate_descent with (l1_reg, l2_reg, add_bias, tol, max_iter,
 * positive = false) -- also when l1_reg <= 0, where `pl_lr`'s dispatch would take the closed-form ridge instead (the two
 * differ: coordinate descent penalises with n_rows * l2_reg, :478-480).  max_iter is honoured in both precisions (the 2000
 * of the f32 expression twin is the plugin function's, not the solver's).  Only an empty frame is rejected (lr/mod.rs:114-125).
 */
int pds_elastic_net_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows, pds_space space,
                        const pds_lr_params* prm, double* coeffs);
int pds_elastic_net_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows, pds_space space,
                        const pds_lr_params* prm, float* coeffs);

/*
 * pds_lr_nullable_*: `pl_lr` / `pl_lr_pred` on columns that carry Arrow validity bitmaps, i.e. the null handling of
 * series_to_mat_for_lr (linear_regression.rs:151-267) done on the device (stream compaction / fill), then the same
 * fit.  validity[c] is the validity bitmap of cols[c] (LSB-first, one bit per row, NULL = column has no nulls),
 * bit_offsets[c] the Arrow array offset into it; both in the same memory space as the column buffers.
 *   pred / resid / row_valid  optional (all NULL = coefficients only), n_rows entries, `space`-resident.  Dropped rows
 *                             get NaN and row_valid = 0 -- the null re-expansion of linear_regression.rs:790-812.
 *   n_used                    out, rows that took part in the fit.
 * Errors: PDS_ERR_NULLS ("Nulls found in data") under PDS_NULL_RAISE; PDS_ERR_EMPTY / PDS_ERR_TOO_FEW_ROWS are
 * judged on the rows that survive the policy, like the reference (:250-254).
 */
int pds_lr_nullable_f64(pds_ctx* ctx, const double* const* cols, const uint8_t* const* validity,
                        const int64_t* bit_offsets, int n_feat, int64_t n_rows, pds_space space, int null_policy,
                        double fill_value, const pds_lr_params* prm, double* coeffs, int* is_null, double* pred,
                        double* resid, uint8_t* row_valid, int64_t* n_used);
int pds_lr_nullable_f32(pds_ctx* ctx, const float* const* cols, const uint8_t* const* validity,
                        const int64_t* bit_offsets, int n_feat, int64_t n_rows, pds_space space, int null_policy,
                        float fill_value, const pds_lr_params* prm, float* coeffs, int* is_null, float* pred,
                        float* resid, uint8_t* row_valid, int64_t* n_used);

/*
 * pds_lr_multi_*: `pl_lr_multi` / `pl_lr_multi_pred` (linear_regression.rs:517-649): k targets share one design
 * matrix.  cols = [t_0 .. t_{k-1}, x_1 .. x_p] (the expression's input order, MultiLRKwargs.last_target_idx = k).
 * OLS / ridge only; the rank gate depends on X alone, so either every target gets coefficients or none does.
 * One pass over the data builds X'X and all k X't_i together (the extra targets ride in the Gram tile).
 *   coeffs  out, k x (n_feat + add_bias), row-major (target i's coefficients contiguous).
 *   pred / resid  optional, k x n_rows row-major, `space`-resident ("{target}_pred" / "{target}_resid" fields).
 */
int pds_lr_multi_f64(pds_ctx* ctx, const double* const* cols, int n_targets, int n_feat, int64_t n_rows,
                     pds_space space, int add_bias, double l2_reg, int solver, double singular_x_tol, double* coeffs,
                     int* is_null, double* pred, double* resid);
int pds_lr_multi_f32(pds_ctx* ctx, const float* const* cols, int n_targets, int n_feat, int64_t n_rows,
                     pds_space space, int add_bias, float l2_reg, int solver, float singular_x_tol, float* coeffs,
                     int* is_null, float* pred, float* resid);

/*
 * pds_lr_rcond_*: `pl_lr_w_rcond` -> faer_solve_lr_rcond (lr_solvers.rs:216-258).
 * rcond is max(kwargs.tol, eps * max(n, p')) as computed by the caller (linear_regression.rs:651-702).
 */
int pds_lr_rcond_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows,
                     pds_space space, int add_bias, double l2_reg, double rcond, double* coeffs,
                     double* singular_values);
/* pl_lr_w_rcond_f32 (linear_regression_f32.rs:515-566): rcond = max(tol as f32, f32::EPSILON * max(n, p')). */
int pds_lr_rcond_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows,
                     pds_space space, int add_bias, float l2_reg, float rcond, float* coeffs,
                     float* singular_values);

/*
 * pds_lin_reg_report_*: the arithmetic of `pl_lin_reg_report` (linear_regression.rs:822-980) and, with
 * weights != NULL, `pl_wls_report` (:982-1117).  y_var is inputs[0][0] of the expression
 * (`target.var()`, ddof = 1, computed by Polars; expr_linear.py:614-617); with se_type | PDS_REPORT_DERIVE_YVAR (unweighted
 * form) the library computes it itself (sum y and sum y^2 in f64 from the rows the report works on).  Each output has
 * n_feat + add_bias entries; r2 / adj_r2 are scalars (the reference broadcasts them).
 */
typedef struct {
    double* beta;
    double* std_err;
    double* t;
    double* p;
    double* ci_lower; /* "0.025" */
    double* ci_upper; /* "0.975" */
    double r2;
    double adj_r2;
} pds_report_f64;
typedef struct {
    float* beta;
    float* std_err;
    float* t;
    float* p;
    float* ci_lower;
    float* ci_upper;
    float r2;
    float adj_r2;
} pds_report_f32;
int pds_lin_reg_report_f64(pds_ctx* ctx, const double* const* cols, const double* weights,
                           int n_feat, int64_t n_rows, pds_space space, int add_bias, int se_type,
                           double y_var, pds_report_f64* out);
int pds_lin_reg_report_f32(pds_ctx* ctx, const float* const* cols, const float* weights,
                           int n_feat, int64_t n_rows, pds_space space, int add_bias, int se_type,
                           float y_var, pds_report_f32* out);

/*
 * Row-sharded `pl_lin_reg_report` / `pl_wls_report` (one process per GPU): the stages of
 * pds_lin_reg_report_* as separate calls, the two exchange steps between them left to the caller.
 *   1. every rank: pds_moments_* on its rows                     -> all-reduce(SUM) of the (p+2)^2 block
 *   2. pds_report_fit_from_moments_*: X'X -> col_piv_qr -> inverse, beta (linear_regression.rs:854-858) -- replicated,
 *      deterministic, so every rank holds the same beta / inv without a broadcast.  moments / beta / inv are HOST buffers
 *      (beta: p' values; inv: p'^2, column-major).
 *   3. every rank: pds_report_partials_* on its rows: the residual pass (:863-909)
 *      partials[0] = sum e^2, partials[1] = sum w e^2 (WLS), partials[2 ..] = the (p+2)^2 block whose leading p' x p' part is
 *      the HC meat X' diag(s) X (zeros for plain standard errors); 2 + (p+2)^2 doubles, HOST -> all-reduce(SUM)
 *   4. pds_report_finish_*: the O(p'^2) epilogue (:861-939) on the summed partials; n_rows_total = rows of the whole frame;
 *      y_var as in pds_lin_reg_report_* (of the whole target).  No device work.
 */
int pds_report_fit_from_moments_f64(pds_ctx* ctx, const double* moments, int n_feat, int add_bias, double* beta, double* inv);
int pds_report_fit_from_moments_f32(pds_ctx* ctx, const float* moments, int n_feat, int add_bias, float* beta, float* inv);
int pds_report_partials_f64(pds_ctx* ctx, const double* const* cols, const double* weights, int n_feat, int64_t n_rows,
                            pds_space space, int add_bias, int se_type, const double* beta, const double* inv, double* partials);
int pds_report_partials_f32(pds_ctx* ctx, const float* const* cols, const float* weights, int n_feat, int64_t n_rows,
                            pds_space space, int add_bias, int se_type, const float* beta, const float* inv, double* partials);
int pds_report_finish_f64(int n_feat, int add_bias, int se_type, int weighted, int64_t n_rows_total, double y_var,
                          const double* beta, const double* inv, const double* partials, pds_report_f64* out);
int pds_report_finish_f32(int n_feat, int add_bias, int se_type, int weighted, int64_t n_rows_total, float y_var,
                          const float* beta, const float* inv, const double* partials, pds_report_f32* out);

/*
 * pds_lin_reg_report_nullable_*: `pl_lin_reg_report` on columns with Arrow validity bitmaps: the null policy of
 * series_to_mat_for_lr (linear_regression.rs:151-267, called at :846) on the device, then the same report on the rows
 * that survive it (dof, r2 use that row count).  Arguments as pds_lr_nullable_*; `y_var` stays what Polars computed
 * on the original target.  `pl_wls_report` has no nullable form: the reference does not compact its weights.
 */
int pds_lin_reg_report_nullable_f64(pds_ctx* ctx, const double* const* cols, const uint8_t* const* validity,
                                    const int64_t* bit_offsets, int n_feat, int64_t n_rows, pds_space space,
                                    int null_policy, double fill_value, int add_bias, int se_type, double y_var,
                                    pds_report_f64* out, int64_t* n_used);
int pds_lin_reg_report_nullable_f32(pds_ctx* ctx, const float* const* cols, const uint8_t* const* validity,
                                    const int64_t* bit_offsets, int n_feat, int64_t n_rows, pds_space space,
                                    int null_policy, float fill_value, int add_bias, int se_type, float y_var,
                                    pds_report_f32* out, int64_t* n_used);

/*
 * pds_lr_grouped_*: the key-aware batched symbol of SURVEY.md 8(b) ("pl_lr_by"): what
 * `df.group_by(key).agg(pds.lin_reg(...))` makes Polars compute by calling `pl_lr` once per group
 * (tests/test_linear_exprs.py:918-953).  Rows of one group are contiguous; group g is rows
 * [group_offsets[g], group_offsets[g+1]).  The dispatch on (l1_reg, l2_reg, positive) is pl_lr's (:447-497), per group:
 * OLS / ridge with the rank gate (the default path; streaming Gram + solve in one kernel), lasso / elastic net /
 * positive fits by coordinate descent on the group's Gram matrix (faer_coordinate_descent, faer_nn_lr).
 *   group_offsets  n_groups + 1 int64 values, `space`-resident.
 *   coeffs         out, n_groups x (n_feat + add_bias), row-major, `space`-resident.
 *   is_null        out, n_groups bytes, `space`-resident: 1 = gated or fewer rows than features
 *                  (per-group `pl_lr` raises / returns null there).
 */
int pds_lr_grouped_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows,
                       const int64_t* group_offsets, int64_t n_groups, pds_space space,
                       const pds_lr_params* prm, double* coeffs, uint8_t* is_null);
int pds_lr_grouped_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows,
                       const int64_t* group_offsets, int64_t n_groups, pds_space space,
                       const pds_lr_params* prm, float* coeffs, uint8_t* is_null);

/*
 * pds_lr_grouped_nullable_*: the same with Arrow validity bitmaps (arguments as pds_lr_nullable_*): what Polars gets
 * from `group_by(key).agg(pds.lin_reg(..., null_policy=...))` -- every group is fitted on the rows of it that survive the
 * policy; a group left with fewer rows than coefficients is null.  PDS_NULL_RAISE fails on the first null anywhere.
 */
int pds_lr_grouped_nullable_f64(pds_ctx* ctx, const double* const* cols, const uint8_t* const* validity,
                                const int64_t* bit_offsets, int n_feat, int64_t n_rows,
                                const int64_t* group_offsets, int64_t n_groups, pds_space space, int null_policy,
                                double fill_value, const pds_lr_params* prm, double* coeffs, uint8_t* is_null);
int pds_lr_grouped_nullable_f32(pds_ctx* ctx, const float* const* cols, const uint8_t* const* validity,
                                const int64_t* bit_offsets, int n_feat, int64_t n_rows,
                                const int64_t* group_offsets, int64_t n_groups, pds_space space, int null_policy,
                                float fill_value, const pds_lr_params* prm, float* coeffs, uint8_t* is_null);

/*
 * pds_lr_with_inv_*: faer_qr_lr_with_inv (lr_online_solvers.rs:120-143) -- the initial fit of OnlineLR (src/pymodels/py_lr.rs
 * :169, lr_online_solvers.rs:101-113): X'X (+ lambda on the n_feat feature diagonals), column-pivoted QR, its inverse and
 * the solution.  coeffs: n_feat + add_bias values (bias last); inv: (n_feat + add_bias)^2 values, column-major (symmetric).
 * Host outputs.  The per-row `woodbury_step` updates that follow are O(p'^2) host arithmetic on this state.
 */
int pds_lr_with_inv_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows, pds_space space, int add_bias,
                        double lambda, double* coeffs, double* inv);
int pds_lr_with_inv_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows, pds_space space, int add_bias,
                        float lambda, float* coeffs, float* inv);

/*
 * pds_lr_grouped_weighted_*: `group_by(key).agg(pds.lin_reg(..., weights=w))` -- per group faer_weighted_lr
 * (lr_solvers.rs:386-409: X'WX, X'Wy, plain solve with `solver`; no gate, no penalties, as pl_lr :436-446).  The frame is
 * scaled by sqrt(w) once on the device and takes the grouped path.  weights: n_rows values, `space`-resident; the other
 * arguments as pds_lr_grouped_*.
 */
int pds_lr_grouped_weighted_f64(pds_ctx* ctx, const double* const* cols, const double* weights, int n_feat, int64_t n_rows,
                                const int64_t* group_offsets, int64_t n_groups, pds_space space, const pds_lr_params* prm,
                                double* coeffs, uint8_t* is_null);
int pds_lr_grouped_weighted_f32(pds_ctx* ctx, const float* const* cols, const float* weights, int n_feat, int64_t n_rows,
                                const int64_t* group_offsets, int64_t n_groups, pds_space space, const pds_lr_params* prm,
                                float* coeffs, uint8_t* is_null);

/*
 * pds_lr_by_key_*: `df.group_by(key).agg(pds.lin_reg(...))` for an int64 key column in ANY row order -- the grouping
 * Polars does on the host before it calls `pl_lr` per group (tests/test_linear_exprs.py:435-474), done on the device:
 * keys already non-decreasing -> no data movement; otherwise a radix sort of (key, row) pairs and a gather of the columns.
 * Then exactly pds_lr_grouped_*.  Groups come back in ascending key order.
 *   keys        n_rows int64 values, `space`-resident.
 *   max_groups  capacity of the outputs (n_rows is always enough).
 *   out_keys    out, the distinct keys;  coeffs  out, n_groups x (n_feat + add_bias);  is_null  out, n_groups bytes
 *               -- all `space`-resident.   n_groups  out (host), number of distinct keys.
 */
int pds_lr_by_key_f64(pds_ctx* ctx, const double* const* cols, const int64_t* keys, int n_feat, int64_t n_rows, pds_space space,
                      const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, double* coeffs, uint8_t* is_null,
                      int64_t* n_groups);
int pds_lr_by_key_f32(pds_ctx* ctx, const float* const* cols, const int64_t* keys, int n_feat, int64_t n_rows, pds_space space,
                      const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, float* coeffs, uint8_t* is_null,
                      int64_t* n_groups);

/*
 * pds_lr_by_key_multi_*: pds_lr_by_key_* for a HOST frame with non-decreasing keys, driven through SEVERAL contexts from one
 * process (the Polars plugin is loaded in-process: python/polars_ds/_utils.py:28-38 -- a torch.distributed launcher cannot
 * serve `pl_lr_by`; SURVEY.md 8(e) "each GPU pulls its own shard over its own PCIe link").  The frame is cut into n_slices row
 * ranges at group boundaries (n_slices <= 0: four per context; a slice has at least 2^20 rows), slice s is fitted by context
 * s mod n_ctx on that context's device and stream from its own host thread, and its keys / coefficients / null flags land
 * directly in the piece of the outputs that follows the groups of the slices in front of it.  Contexts on DIFFERENT devices:
 * every device pulls its shard over its own link.  Several contexts on ONE device: a slice's transfer overlaps the previous
 * slice's fit and result copy.  Results are identical to pds_lr_by_key_* (same kernels per group).
 * Keys that are NOT in order (round 4; SURVEY.md 8(e) row C3 "hash(key) % R", here without moving rows): one row slice per context
 * whatever the order; every context builds the id-indexed moment table of its rows (dense integer keys, <= 16 features: the
 * partition route of pds_lr_by_key_*), the tables are summed on ctxs[0] (same device: in place; another device: hipMemcpyPeer),
 * which lists the groups and solves them -- one exchange of (key range) x (p+2)(p+3)/2 doubles per extra context.  Frames the
 * partition route does not take (sparse keys, wider frames, the "keyed_sort" option of ctxs[0]) and n_ctx == 1 take pds_lr_by_key_* on ctxs[0].
 * Arguments as pds_lr_by_key_* (space = PDS_HOST).
 */
int pds_lr_by_key_multi_f64(pds_ctx* const* ctxs, int n_ctx, int n_slices, const double* const* cols, const int64_t* keys, int n_feat,
                            int64_t n_rows, const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, double* coeffs,
                            uint8_t* is_null, int64_t* n_groups);
int pds_lr_by_key_multi_f32(pds_ctx* const* ctxs, int n_ctx, int n_slices, const float* const* cols, const int64_t* keys, int n_feat,
                            int64_t n_rows, const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, float* coeffs,
                            uint8_t* is_null, int64_t* n_groups);
/*
 * pds_lr_by_key_pred_multi_*: the per-row predictions of pds_lr_by_key_pred_* (pred / resid / row_null of the frame's length, in the
 * frame's row order; each nullable, not all) for a HOST frame with non-decreasing keys over several contexts -- what
 * `pds.lin_reg(..., return_pred=True).over(key)` (tests/test_linear_exprs.py:435-474) needs for a host frame: slice s + 1 crosses
 * the link while slice s is fitted and its predictions travel back (PCIe is full duplex), and contexts on different devices use
 * their own links.  Slices are independent (no group list comes back).  Keys not in order: pds_lr_by_key_pred_* on ctxs[0].
 *   weights  nullable, n_rows values (per-group faer_weighted_lr).
 */
int pds_lr_by_key_pred_multi_f64(pds_ctx* const* ctxs, int n_ctx, int n_slices, const double* const* cols, const double* weights,
                                 const int64_t* keys, int n_feat, int64_t n_rows, const pds_lr_params* prm, double* pred, double* resid,
                                 uint8_t* row_null);
int pds_lr_by_key_pred_multi_f32(pds_ctx* const* ctxs, int n_ctx, int n_slices, const float* const* cols, const float* weights,
                                 const int64_t* keys, int n_feat, int64_t n_rows, const pds_lr_params* prm, float* pred, float* resid,
                                 uint8_t* row_null);
/*
 * The three exchange steps of SURVEY.md 8(e) for a host that drives SEVERAL devices from ONE process (a Rust plugin inside the
 * Polars process: no launcher, no RCCL communicator) -- peer copies between the contexts' devices plus one kernel, each ordered on
 * the contexts' own streams; every call returns after all of its contexts' streams have been synchronised.  A host with one
 * process per device uses its collective library for the same three steps (python: polars_ds_extension_amd/parallel.py over
 * torch.distributed / RCCL); the compute entry points on either side are the same (pds_moments_*, pds_lr_from_moments_*,
 * pds_report_partials_*, pds_recursive_lr_seeded_*, pds_lr_grouped_*).
 *   pds_allreduce_sum_f64   bufs[c]: `count` doubles resident on ctxs[c]'s device (moment blocks, report partials, the seeds of the
 *                           expanding fit); afterwards every buffer holds the element-wise sum in rank order 0, 1, ..: rows C2 /
 *                           C5 ("all-reduce of the Gram block") and C4 ("prefix Gram": with `prefix != 0` buffer c receives the
 *                           sum of buffers 0 .. c-1 instead -- the exclusive scan the row-sharded expanding fit seeds with).
 *   pds_scatter_rows_f64    a frame resident on ctxs[0]'s device: column k's rows [bounds[c], bounds[c+1]) -> dst[c][k] on ctxs[c]'s
 *                           device (row C3, device-resident frame: "root -> peers, all links at once"); dst[0] may be null (rank
 *                           0 keeps reading the frame in place).
 *   pds_gather_f64          src[c]: counts[c] doubles on ctxs[c]'s device -> dst on ctxs[0]'s device, back to back in rank order
 *                           (row C3: the coefficient blocks of the ranks' groups).
 * f32 twins move floats.  Their all-reduce keeps the running sum as a float: every single addition is formed in f64 and rounded back,
 * so n_ctx addends cost up to n_ctx - 1 float roundings -- not an f64 accumulation (moment blocks that need one travel as f64:
 * pds_moments_* returns f64 records for both precisions).
 */
int pds_allreduce_sum_f64(pds_ctx* const* ctxs, int n_ctx, double* const* bufs, int64_t count, int prefix);
int pds_allreduce_sum_f32(pds_ctx* const* ctxs, int n_ctx, float* const* bufs, int64_t count, int prefix);
int pds_scatter_rows_f64(pds_ctx* const* ctxs, int n_ctx, const double* const* cols, int n_cols, const int64_t* bounds,
                         double* const* const* dst);
int pds_scatter_rows_f32(pds_ctx* const* ctxs, int n_ctx, const float* const* cols, int n_cols, const int64_t* bounds,
                         float* const* const* dst);
int pds_gather_f64(pds_ctx* const* ctxs, int n_ctx, const double* const* src, const int64_t* counts, double* dst);
int pds_gather_f32(pds_ctx* const* ctxs, int n_ctx, const float* const* src, const int64_t* counts, float* dst);

/* Page-locked host storage (hipHostMalloc, portable across devices) for result buffers the caller owns: device-to-host copies
 * into it run at the link rate.  The plugin layer keeps the large Arrow result buffers in such storage (plugin_arrow_out.hpp). */
int pds_host_alloc(size_t bytes, void** out);
int pds_host_free(void* p);
/* number of visible devices (the plugin layer's PDS_DEVICES=all) */
int pds_device_count(int* n);
/*
 * Result storage another PROCESS's kernels can write (one process per device, SURVEY.md 8(e) row C3 "results: gather of n_groups_r x p'
 * f64 + validity to rank 0"): the gathering rank allocates the assembled [n_groups][p'] block with pds_device_alloc (hipMalloc on
 * `device`: an IPC handle names a whole allocation), exports it, hands the 64 handle bytes to its peers (any side channel: the
 * launcher's store, a broadcast), every peer maps it with pds_ipc_open on ITS device (peer access over xGMI is enabled by the
 * mapping) and passes `mapped + its groups' offset` as the `coeffs` / `is_null` arguments of pds_lr_grouped_*: the fit's stores
 * cross the link while the kernel runs -- no staging buffer, no send / receive launches, no separate gather step.  The gathering rank
 * may read the rows once the peer's stream has passed the call (completion: below).
 * pds_ipc_close unmaps (the allocation stays the exporter's, freed with pds_device_free after every peer has closed).
 * python: polars_ds_extension_amd/parallel.py, GroupedShardPlan(direct=True).
 *
 * Completion without a collective: a SIGNAL block (pds_device_alloc with fine_grained = 1: device memory that is coherent across
 * devices, as RCCL's own flag words are) shared the same way holds one 32-bit word per rank.  A peer calls
 * pds_signal_post(ctx, mapped_word, seq) behind its fit -- a one-lane kernel on the context's stream: the fit's stores are complete and
 * visible at the kernel boundary before it, the word is stored with system scope --; the gathering rank calls
 * pds_signal_wait(ctx, words, n, seq) on ITS stream: a one-wave kernel that polls the n words (system-scope loads) until each is
 * >= seq.  Both are stream ordered: no host synchronisation, no collective launch; a step is the fit + one tiny kernel on every rank.
 * The wait gives up after `timeout_ms` (a peer that died must not hang the device): the call then fails at the next synchronisation
 * point through pds_signal_wait_status (0 = every wait so far was satisfied).  fine_grained = 1 also for the RESULT block of a direct
 * gather: the gathering device's L2 must not keep lines of rows another device writes.
 */
#define PDS_IPC_HANDLE_BYTES 64
int pds_device_alloc(int device, size_t bytes, int fine_grained, void** out);
int pds_signal_post(pds_ctx* ctx, unsigned* word, unsigned value);
int pds_signal_wait(pds_ctx* ctx, const unsigned* words, int n_words, unsigned value, int timeout_ms);
int pds_signal_wait_status(pds_ctx* ctx, int* timed_out);
int pds_device_free(int device, void* p);
int pds_ipc_export(int device, void* p, unsigned char* handle);             /* handle: PDS_IPC_HANDLE_BYTES bytes, written */
int pds_ipc_open(int device, const unsigned char* handle, void** mapped);  /* handle: PDS_IPC_HANDLE_BYTES bytes */
int pds_ipc_close(int device, void* mapped);

/*
 * pds_lr_grouped_pred_* / pds_lr_by_key_pred_*: the grouped form of `pl_lr_pred` (linear_regression.rs:704-820) -- what
 * `df.group_by(key).agg(pds.lin_reg(..., return_pred=True))` (tests/test_linear_exprs.py:435-474) and
 * `pds.lin_reg(..., return_pred=True).over(key)` make Polars compute one group at a time: every group is fitted as
 * pds_lr_grouped_* / pds_lr_by_key_* fit it, then ONE more pass over the frame (grouped_pred.hip) writes, for every row,
 *   pred[r] = x_r . beta_g(r) (+ intercept),  resid[r] = y_r - pred[r],  row_null[r] = 1 when the row's group is null
 * (pred / resid are NaN there; the reference returns an all-null struct for such a group, :745-750) -- IN THE FRAME'S OWN ROW
 * ORDER, also when the keys are shuffled.  weights (nullable, n_rows values): per group faer_weighted_lr as in
 * pds_lr_grouped_weighted_*, predictions from the unweighted rows.  pred / resid / row_null are `space`-resident and each may
 * be NULL (at least one is not); coeffs / is_null (grouped) and out_keys / coeffs / is_null / n_groups (by key) may be NULL
 * when only the per-row outputs are wanted (max_groups is then ignored).  Null-free frames only.
 */
int pds_lr_grouped_pred_f64(pds_ctx* ctx, const double* const* cols, const double* weights, int n_feat, int64_t n_rows,
                            const int64_t* group_offsets, int64_t n_groups, pds_space space, const pds_lr_params* prm, double* coeffs,
                            uint8_t* is_null, double* pred, double* resid, uint8_t* row_null);
int pds_lr_grouped_pred_f32(pds_ctx* ctx, const float* const* cols, const float* weights, int n_feat, int64_t n_rows,
                            const int64_t* group_offsets, int64_t n_groups, pds_space space, const pds_lr_params* prm, float* coeffs,
                            uint8_t* is_null, float* pred, float* resid, uint8_t* row_null);
int pds_lr_by_key_pred_f64(pds_ctx* ctx, const double* const* cols, const double* weights, const int64_t* keys, int n_feat,
                           int64_t n_rows, pds_space space, const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys,
                           double* coeffs, uint8_t* is_null, int64_t* n_groups, double* pred, double* resid, uint8_t* row_null);
int pds_lr_by_key_pred_f32(pds_ctx* ctx, const float* const* cols, const float* weights, const int64_t* keys, int n_feat, int64_t n_rows,
                           pds_space space, const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, float* coeffs,
                           uint8_t* is_null, int64_t* n_groups, float* pred, float* resid, uint8_t* row_null);

/*
 * pds_rolling_lr_* / pds_recursive_lr_*: `pl_rolling_lr` (linear_regression.rs:1206-1283) and
 * `pl_recursive_lr` (:1121-1204) on null-free columns, i.e. faer_rolling_lr / faer_recursive_lr
 * (lr_online_solvers.rs:148-212) plus the plugin's pred_i = x_i . coeffs_i.  SWWLRKwargs: n = window
 * (or start_with), bias, lambda.  As in the reference the ones column is part of the data and lambda is
 * added to every diagonal entry.
 *   coeffs  out, n_rows x (n_feat + add_bias) row-major (the values buffer of the List column),
 *           `space`-resident; rows [0, n-1) are unspecified and marked invalid.
 *   pred    out, n_rows, `space`-resident.
 *   valid   out, n_rows bytes (1 = row has a result), `space`-resident.
 * Coefficients (n_feat + add_bias): <= 8 lane = 4 rows, 9 .. 12 lane = row (one kernel each), 13 .. 64 through per-row
 * moment records and the batched pivoted QR, 65 .. 255 (n_feat <= 254) through a 1024-thread record kernel and the big-system
 * solver (coverage path: the reference's drivers have no limit).
 * min_size > 0 selects the skipping variant (faer_rolling_skipping_lr :218-301): rows holding a
 * non-finite value are left out of the window and a window with fewer than min_size rows is invalid.
 */
int pds_rolling_lr_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows,
                       pds_space space, int add_bias, int64_t window, int64_t min_size,
                       double lambda, double* coeffs, double* pred, uint8_t* valid);
int pds_rolling_lr_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows,
                       pds_space space, int add_bias, int64_t window, int64_t min_size,
                       float lambda, float* coeffs, float* pred, uint8_t* valid);
int pds_recursive_lr_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows,
                         pds_space space, int add_bias, int64_t start_with, double lambda,
                         double* coeffs, double* pred, uint8_t* valid);
int pds_recursive_lr_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows,
                         pds_space space, int add_bias, int64_t start_with, float lambda,
                         float* coeffs, float* pred, uint8_t* valid);

/*
 * pds_recursive_lr_seeded_*: the expanding fit of a frame that CONTINUES earlier rows -- the row-sharded
 * multi-GPU form of `pl_recursive_lr` ("recursive needs prefix Gram").  `seed_moments` is the
 * HOST-resident augmented moment matrix A = Z'Z ((n_feat+2)^2, layout of pds_moments_*) of all rows in
 * front of this frame (rank r passes the sum of the moment matrices of ranks < r; NULL = no rows = the
 * plain call).  The fit at local row i is over seed + rows [0, i]; the seed's row count (entry [n_feat, n_feat]) counts
 * towards start_with.  The seed must come from finite rows only.
 */
int pds_recursive_lr_seeded_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows,
                                pds_space space, int add_bias, int64_t start_with, double lambda,
                                const double* seed_moments, double* coeffs, double* pred,
                                uint8_t* valid);
int pds_recursive_lr_seeded_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows,
                                pds_space space, int add_bias, int64_t start_with, float lambda,
                                const float* seed_moments, float* coeffs, float* pred,
                                uint8_t* valid);

/*
 * Moment-level entry points (what a multi-GPU host composes; also the measured Gram build).
 *
 * pds_moments_*: one streaming pass over [y | X] building the augmented moment matrix
 *     A = Z'Z,  Z = [x1 .. xp | 1 | y]   ((p+2) x (p+2), column-major, symmetric)
 * i.e. X'X (get_xtx_with_lambda lr_solvers.rs:183-198), X'y (build_xty :262-278), the column sums and
 * sum(y) used by the coordinate-descent bias shortcut (:483-484), y'y and n -- every input element is
 * read from HBM exactly once.  With weights != NULL: Z'WZ.  The result is written to `moments`
 * ((n_feat+2)^2 values) in `out_space`; for PDS_DEVICE the call only enqueues work (no host sync), so
 * a caller can all-reduce the buffer across ranks (RCCL) and then call pds_lr_from_moments_*.
 */
int pds_moments_f64(pds_ctx* ctx, const double* const* cols, const double* weights, int n_feat,
                    int64_t n_rows, pds_space space, double* moments, pds_space out_space);
int pds_moments_f32(pds_ctx* ctx, const float* const* cols, const float* weights, int n_feat,
                    int64_t n_rows, pds_space space, float* moments, pds_space out_space);
/* Same dispatch as pds_lr_* but starting from (already all-reduced) moments in `mom_space`. */
int pds_lr_from_moments_f64(pds_ctx* ctx, const double* moments, pds_space mom_space, int n_feat,
                            const pds_lr_params* prm, double* coeffs, int* is_null);
int pds_lr_from_moments_f32(pds_ctx* ctx, const float* moments, pds_space mom_space, int n_feat,
                            const pds_lr_params* prm, float* coeffs, int* is_null);

/* Bit-faithful host restatements of src/stats_utils (beta.rs:24-37, 365-377) used by the report. */
double pds_student_t_sf(double x, double df);
double pds_student_t_ppf(double q, double df);

/* ------------------------------------------------------------------------------------------------
 * The pyclass route (src/pymodels): GLM, fits from row-major matrices
 * ---------------------------------------------------------------------------------------------- */
/* GLM by iteratively re-weighted least squares -- faer_irls (src/linear/glm/glm_solvers.rs:249-368) as GLM::fit_unchecked
 * drives it (:216-240: weighted least squares by pivoted QR, a ones column for add_bias).  link: 0 identity, 1 log, 2 logit,
 * 3 inverse; variance: 0 gaussian, 1 poisson, 2 binomial, 3 gamma (link_functions.rs:5-77; GLMFamily::link_function /
 * variance_function glm_solvers.rs:24-41).  coeffs: n_feat + add_bias values, bias last; *n_iter (nullable): iterations run.
 * Stops when max |beta_new - beta| < tol or after max_iter iterations (the reference does not report non-convergence).
 * Up to 16 feature columns an iteration is ONE pass over the frame (weights and working response formed inside the Gram
 * kernel); wider frames write them as two columns and run the weighted wide Gram build on them. */
int pds_glm_irls_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows, pds_space space, int add_bias, int link,
                     int variance, double tol, int max_iter, double* coeffs, int* n_iter);
int pds_glm_irls_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows, pds_space space, int add_bias, int link,
                     int variance, float tol, int max_iter, float* coeffs, int* n_iter);

/* Fits straight from a ROW-MAJOR matrix -- the pyclass route (PyLR / PyElasticNet / PyOnlineLR, src/pymodels/py_lr.rs:21-224,
 * whose NumPy X the reference reads through a strided faer MatRef, src/pymodels/numpy_faer.rs:10-66).
 * X: n_rows x n_feat values, row stride ld (elements); y: n_rows values; both resident in `space`.
 * mode 0: LR::fit -- the pl_lr dispatch of `prm` (faer_solve_lr, lr_solvers.rs:65-73; pass singular_x_tol = 0 for the model class);
 * mode 1: ElasticNet::fit -- always faer_coordinate_descent with prm's l1_reg / l2_reg / tol / max_iter (lr_solvers.rs:139-164);
 * mode 2: OnlineLR::fit -- faer_qr_lr_with_inv (lr_online_solvers.rs:120-143): coeffs and inv = (X'X + lambda I)^-1 with
 *         lambda = prm->l2_reg on the feature diagonals, (n_feat + add_bias)^2 values column-major (inv required).
 * Up to 16 features the matrix core reads the rows as they lie (one pass, no transposition; host matrices as contiguous row
 * chunks); wider matrices are transposed once on the device.  coeffs: n_feat + add_bias values, bias last. */
int pds_lr_rowmajor_f64(pds_ctx* ctx, const double* X, int64_t ld, const double* y, int64_t n_rows, int n_feat, pds_space space,
                        const pds_lr_params* prm, int mode, double* coeffs, int* is_null, double* inv);
int pds_lr_rowmajor_f32(pds_ctx* ctx, const float* X, int64_t ld, const float* y, int64_t n_rows, int n_feat, pds_space space,
                        const pds_lr_params* prm, int mode, float* coeffs, int* is_null, float* inv);

/* Row-major matrix -> the contiguous column buffers every entry point takes (the pyclass route: the NumPy X of LR /
 * ElasticNet / OnlineLR, which the reference reads through a strided faer MatRef, src/pymodels/numpy_faer.rs:10-66).
 * X: n_rows x n_cols values with row stride ld (elements), resident in `space`.  out_cols: DEVICE buffer; column c is
 * written to out_cols + c * col_stride (col_stride >= n_rows).  Host matrices cross PCIe as contiguous row chunks. */
int pds_rows_to_cols_f64(pds_ctx* ctx, const double* X, int64_t ld, int64_t n_rows, int n_cols, pds_space space, double* out_cols,
                         int64_t col_stride);
int pds_rows_to_cols_f32(pds_ctx* ctx, const float* X, int64_t ld, int64_t n_rows, int n_cols, pds_space space, float* out_cols,
                         int64_t col_stride);

#ifdef __cplusplus
}
#endif
#endif /* PDS_LSTSQ_H */
