// common.hpp -- shared declarations of libpds_lstsq_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pds_lstsq.h"

// Wave-private LDS hand-off (one lane writes, another lane of the SAME wave reads): LDS operations of a wave
// execute in issue order, so all that is needed is (a) the compiler not moving LDS accesses across this point and
// (b) outstanding LDS traffic drained.  A `fence(acq_rel, "wavefront")` also works but makes the compiler wait for
// EVERY outstanding memory operation (vmcnt(0)) -- which silently serialised the register prefetch of the next tile
// behind every LDS hand-off in the streaming kernels.
#define PDS_WAVE_LDS_SYNC()                                   \
    do {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_wave_barrier();                      \
    } while (0)

// Newton steps x <- x + x (1 - d x) behind v_rcp_f64 for the pivot reciprocals of the in-register L D L' solves
// (solve_reg_dev.hpp, rolling_seg_dev.hpp); tools/rcp_accuracy.hip measures what the instruction delivers on its own and after
// each step
#ifndef PDS_RCP_NEWTON
#define PDS_RCP_NEWTON 2
#endif

namespace pds {

// Column base pointers reach the kernels through a device-side table, so the compiler only knows them as generic
// ("flat") addresses.  Flat loads tick lgkmcnt as well as vmcnt -- every wait for an LDS or scalar result then also
// waits for the HBM loads in flight -- and cannot use the SGPR-base + 32-bit-VGPR-offset addressing form.  The
// kernels therefore retag the pointers as global (address space 1) once, where they fetch them.
#if defined(__HIPCC__)
template <typename T>
using gptr = const __attribute__((address_space(1))) T*;
template <typename T>
__device__ __forceinline__ gptr<T> as_global(const T* p) {
    return (gptr<T>)p;
}
#endif


// ---------------------------------------------------------------------------------------------
// error plumbing: thread-local message, like the plugin ABI's `_polars_plugin_get_last_error_message`
// ---------------------------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define PDS_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return pds::fail(PDS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));  \
    } while (0)

// ---------------------------------------------------------------------------------------------
// moment-matrix layout.  For p features: q = p + 2, Z = [x_0 .. x_{p-1} | 1 | y], A = Z'Z stored
// column-major q x q.  Index helpers:
//   A[i + j*q], i,j < p : X'X          A[i + p*q]      : column sums (bias row/col)
//   A[p + p*q]          : n (or sum w) A[i + (p+1)*q]  : X'y
//   A[p + (p+1)*q]      : sum y        A[(p+1)+(p+1)*q]: y'y
// ---------------------------------------------------------------------------------------------
constexpr int kMaxFeatSmall = 16;   // one MFMA 16x16 block of features
constexpr int kMaxFeatWide = 64;    // four blocks; beyond that -> tiled SYRK kernel (f32) / unsupported
constexpr int kTileBytesPerLane = 16;

// per-block partial record of the small-p moment kernel (doubles):
//   [0,256)  D[i + 16*j]   X'X tile     [256,272) xy[f]   [272,288) cs[f]   288 yy  289 ys  290 sw
constexpr int kPartD = 0, kPartXY = 256, kPartCS = 272, kPartYY = 288, kPartYS = 289, kPartSW = 290;
constexpr int kPartY1 = 291, kPartY2 = 292;  // spare slots: sum (y - y[0]), sum (y - y[0])^2 (IrlsArgs::y_sums)
constexpr int kPartStride = 296;

struct Workspace {
    void* ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace pds

struct pds_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    double* partials = nullptr; // per-block partial records (num_cus * 8 blocks * kPartStride doubles)
    pds::Workspace ws;       // HBM scratch for call-local arrays (bump allocated per call)
    size_t ws_used = 0;
    std::vector<void*> ws_spill;     // slices ws_take() had to allocate on their own (an under-estimated ws_reserve bound)
    long long ws_spill_count = 0;    // ... counted since the context was created (pds_ctx_workspace_spills)
    pds::Workspace stage;    // HBM staging of PDS_HOST column buffers
    pds::Workspace solve_ws; // factor workspace of the p' > 64 solver (solve_big.hip)
    pds::Workspace keyed;    // pds_lr_by_key_*: staged / sorted keys, permutation, gathered columns, run-length results
    pds::Workspace wkeyed;   // pds_lr_grouped_weighted_*: staged + sqrt(w)-scaled frame (separate: by_key calls it with its frame in `keyed`)
    // fused grouped kernel -> pivoted-QR pass hand-over: device counter of marked groups, and a host-mapped word the kernel
    // raises when it marks its first group (the only thing the host reads in the common case)
    unsigned* mark_count = nullptr;
    unsigned* mark_host = nullptr;      // host address
    unsigned* mark_host_dev = nullptr;  // the same word as the device sees it
    int* wait_timeouts = nullptr;       // device counter of pds_signal_wait kernels that gave up (pds_signal_wait_status)
    bool mark_dirty = false;            // a launch went out and mark_count has not been seen back at zero since (an error path)
    void* pinned = nullptr;  // pinned host scratch (small results, pointer arrays)
    size_t pinned_bytes = 0;
    void* pinned_in = nullptr;  // pinned staging of small PDS_HOST frames: all columns + pointer table, ONE H2D copy
    size_t pinned_in_bytes = 0;
    // optional per-kernel-class HIP-event timing (pds_ctx_set_timing / pds_ctx_get_timing)
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    struct EvPair { int kind; hipEvent_t a, b; };
    std::vector<EvPair> ev_pending;
    // behaviour switches of the context (pds_ctx_set_option); their defaults come from the environment once, at pds_ctx_create
    bool opt_keyed_sort = false;       // "keyed_sort" / PDS_KEYED_SORT=1: unordered keys always take the sorting route (determinism)
    bool opt_wide_f32_native = false;  // "wide_f32_native" / PDS_WIDE_F32_NATIVE=1: f32 Gram beyond 64 features on the f32 matrix instructions
    double kind_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long kind_count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<float> kind_samples[8];  // the individual bracketed durations (ms), newest kept up to 4096 per class
};

namespace pds {

// Development switches -- the A/B alternative of a default that won its measurement, timing experiments -- exist only in libraries
// built with EXTRA=-DPDS_DEV_SWITCHES (tools/README.md).  A product build compiles the default in and never looks them up.
#ifdef PDS_DEV_SWITCHES
inline const char* dev_env(const char* name) { return std::getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; }
#endif

// kernel classes for the timing hooks
enum { kKindMoments = 0, kKindGroupedMoments = 1, kKindSolve = 2, kKindPass2 = 3, kKindRolling = 4, kKindIter = 5 };
// RAII: records a HIP event pair around the launches issued in its scope when ctx->timing is on
struct KernelTimer {
    pds_ctx* ctx;
    int kind;
    hipEvent_t a = nullptr, b = nullptr;
    KernelTimer(pds_ctx* c, int k);
    ~KernelTimer();
};

int ensure_ws(pds_ctx* ctx, Workspace& w, size_t bytes);
int ensure_pinned(pds_ctx* ctx, size_t bytes);
// call-local bump allocation inside ctx->ws: ws_reserve() once per API call with an upper bound, then
// ws_take() hands out 256-byte aligned slices; a request past the reserved bound is served from its own allocation and
// counted (never silently out of bounds).
int ws_reserve(pds_ctx* ctx, size_t total_bytes);
void* ws_take(pds_ctx* ctx, size_t bytes);

// Columns resident in HBM for the duration of one call.  For PDS_DEVICE inputs this is just the
// pointer array copied to the device; for PDS_HOST inputs the column buffers are staged into HBM.
template <typename T>
struct DeviceCols {
    const T** d_ptrs = nullptr;  // device array of nc pointers
    std::vector<const T*> h_ptrs; // the same pointers on the host
    int nc = 0;
};

// order on device: x_0..x_{p-1}, y, [w]
template <typename T>
int make_device_cols(pds_ctx* ctx, const T* const* cols /*[y,x1..xp]*/, const T* weights, int n_feat,
                     int64_t n_rows, pds_space space, DeviceCols<T>& out);

// one IRLS step folded into the Gram pass (moments.hip WM = 3): link / variance ids as in pds_lstsq.h, init = first iteration
struct IrlsArgs {
    int link = 0, variance = 0, init = 0;
    double y_mean = 0.0;
    // WM = 4 (HC2 / HC3 meat in the residual pass): (X'X)^-1, p' x p' column-major in the kernel's T, and the power of
    // 1 / (1 - h_i) that scales the squared residual (1: HC2, 2: HC3)
    const void* inv = nullptr;
    int hc_pow = 0;
    // residual-weighted passes of lin_reg_report (WM = 2 / 4): also sum (y - y[0]) and sum (y - y[0])^2 of the rows the pass reads anyway --
    // the target's variance (PDS_REPORT_DERIVE_YVAR) without one more pass over y; they travel in two spare slots of the partial records
    int y_sums = 0;
};

// ---- kernels' host launchers (moments.hip) ----
template <typename T>
int launch_moments(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, bool weighted,
                   T* d_moments /*device, (p+2)^2*/,
                   // p <= 16, unweighted frames only: weight of row i = (y_i - x_i . beta [- beta_p])^2 formed inside the pass
                   // (the HC0 / HC1 meat of lin_reg_report); d_sums_resid[0] receives sum of the weights = sum e^2
                   const T* d_beta_resid = nullptr, int bias_resid = 0, double* d_sums_resid = nullptr,
                   // p <= 16: write the record as f64 into this slot instead of d_moments (row chunks of a host frame)
                   double* d_moments_f64 = nullptr,
                   // p <= 16: weights and working response of one IRLS step from d_beta_resid (moments = X'WX | X'Wz)
                   const IrlsArgs* irls = nullptr,
                   // p <= 16, residual-weighted passes (d_beta_resid): sum (y - y[0]), sum (y - y[0])^2 -> d_ysums[0..1] (see IrlsArgs::y_sums)
                   double* d_ysums = nullptr);
// moments.hip: weights and working response of one IRLS step for frames of more than 16 features (the wide Gram build reads them)
template <typename T>
int launch_irls_working_wide(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, int bias, const T* d_beta,
                             const IrlsArgs& ia, T* d_w, T* d_z);
// moments.hip: Gram build straight from a row-major device matrix (X[r * ld + c], y[r]), 1..16 features, unweighted
template <typename T>
int launch_moments_rowmajor(pds_ctx* ctx, const T* d_X, int64_t ld, const T* d_y, int n_feat, int64_t n_rows, T* d_moments,
                            double* d_moments_f64 = nullptr);
template <typename T>
int launch_sum_moment_slots(pds_ctx* ctx, const double* d_slots, int nslots, int len, T* d_out);

// moments_wide.hip: p > 16 (tiled MFMA SYRK with split-K); partials come out of ctx->ws (reserve
// moments_wide_workspace() bytes on top of the call's other needs)
template <typename T>
int launch_moments_wide(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, bool weighted, T* d_moments);
size_t moments_wide_workspace(int num_cus, int n_feat, int64_t n_rows, bool weighted = false);
// moments_mid.hip: 17 .. 64 f64 features -- the streaming kernel (several 16-feature tile columns); its per-wave
// partial records come out of ctx->ws like the wide kernel's (moments_wide_workspace() covers them)
size_t moments_mid_workspace(int num_cus);
template <typename T>
int launch_moments_mid(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, bool weighted, T* d_moments);

// segmented (per-group) moments: d_moments [n_groups][(p+2)^2].  d_group_index (p <= 16 only): record g belongs to group
// d_group_index[g] of the offsets array instead of group g.
template <typename T>
int launch_grouped_moments(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, const int64_t* d_offsets,
                           int64_t n_groups, T* d_moments, const int32_t* d_group_index = nullptr);

struct SolveParams;
// grouped_fused.hip: per-group Gram + gated Cholesky in one streaming kernel (p <= 16, OLS / ridge); unless the caller asked
// for "choleskey", groups next to the gate go through a second, pivoted-QR pass (d_mom_scratch: room for `scratch_groups`
// moment records of (p+2)^2 values; the pass is chunked by it)
template <typename T>
int launch_grouped_fused(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, const int64_t* d_offsets,
                         int64_t n_groups, const SolveParams& sp, T* d_coeffs, uint8_t* d_flags, T* d_mom_scratch,
                         int64_t scratch_groups);

// ---- solve.hip ----
struct SolveParams {
    int p;         // features (without bias)
    int add_bias;
    int solver;
    double lambda; // l2 added to the first p diagonals (not the bias)
    double gate_tol; // <= 0: no gate
    int lambda_on_bias; // rolling / recursive: lambda on every diagonal
};
// batched: moments [n_sys][(p+2)^2] -> coeffs [n_sys][p+bias], flags [n_sys] (1 = null)
// optional inv_out [n_sys][p'*p'] (column-major) = (X'X + lambda)^-1 via the QR, and logdet.
template <typename T>
int launch_solve(pds_ctx* ctx, const T* d_moments, int64_t n_sys, const SolveParams& sp, T* d_coeffs,
                 uint8_t* d_flags, T* d_inv_out, const int64_t* d_rows_per_sys /*nullable*/);

// capi_lr.hpp: the marked systems of a grouped fit through the reference's factorisation for sp.solver (pivoted QR; "svd": host SVD gate)
template <typename T>
int launch_solve_marked(pds_ctx* ctx, const T* d_rec, int64_t n_sys, const SolveParams& sp, T* d_coeffs, uint8_t* d_flags,
                        const int64_t* d_rows_per_sys);

// solve_big.hip: p' > 64 (Cholesky on an HBM/L2 workspace, one workgroup per system); used by launch_solve
template <typename T>
int launch_solve_big(pds_ctx* ctx, const T* d_moments, int64_t n_sys, const SolveParams& sp, T* d_coeffs, uint8_t* d_flags,
                     T* d_inv_out);

// solve_reg.hip: register-resident variant (p' <= 16, QR, no inverse); used by launch_solve.  `tri` (nullable): the systems are rows
// ids[s] of an id-indexed table of packed upper triangles over [x_0 .. x_{pc-1}, 1, y] (keyed_partition.hip) instead of d_moments
struct TriSource {
    const double* table = nullptr;
    const unsigned* ids = nullptr;
    int nvp = 0, pc = 0;
};
template <typename T>
int launch_solve_reg(pds_ctx* ctx, const T* d_moments, int64_t n_sys, const SolveParams& sp, T* d_coeffs,
                     uint8_t* d_flags, const int64_t* d_rows_per_sys, const TriSource* tri = nullptr);

// solve_wave.hip: 17 .. 64 features, one wave per system, L D L' in registers + a pivoted-QR pass over the systems it marks
// (PDS_ERR_UNSUPPORTED without an error message: not applicable, the caller keeps launch_solve)
template <typename T>
int launch_solve_wave(pds_ctx* ctx, const T* d_moments, int64_t n_sys, const SolveParams& sp, T* d_coeffs, uint8_t* d_flags,
                      const int64_t* d_rows_per_sys, void* d_ws);
size_t solve_wave_workspace(int n_feat, int add_bias, int64_t n_sys, size_t elem);

template <typename T>
int launch_cd(pds_ctx* ctx, const T* d_moments, int p, int add_bias, double l1, double l2, double tol,
              int max_iter, int positive, T* d_coeffs, int* d_info /*per system: [0]=sweeps,[1]=converged; nullable*/,
              int64_t n_sys = 1, uint8_t* d_flags = nullptr, const int64_t* d_rows_per_sys = nullptr);
template <typename T>
int launch_nnls(pds_ctx* ctx, const T* d_moments, int p, int add_bias, double tol, int max_iter, T* d_coeffs,
                int64_t n_sys = 1, uint8_t* d_flags = nullptr, const int64_t* d_rows_per_sys = nullptr);

// ---- grouped_pred.hip: per-row pred / resid of grouped fits (frame in group order; d_perm sends rows back to frame order) ----
template <typename T>
int launch_grouped_pred(pds_ctx* ctx, const T* const* d_cols, int n_feat, int bias, int64_t n_rows, const int64_t* d_off,
                        int64_t n_groups, const T* d_coeffs, const uint8_t* d_flags, const uint32_t* d_perm, T* d_pred, T* d_resid,
                        uint8_t* d_row_null);

template <typename T>
int launch_grouped_pred_by_id(pds_ctx* ctx, const T* const* d_cols, int n_feat, int bias, int64_t n_rows, const int64_t* d_keys,
                              const int64_t* d_kmin, const uint32_t* d_rank, int64_t n_groups, const T* d_coeffs, const uint8_t* d_flags,
                              T* d_pred, T* d_resid, uint8_t* d_row_null);
// elements per row of the id-indexed coefficient table: rows padded to whole 64- or 128-byte lines (wider rows: as they are)
template <typename T>
inline int grouped_pred_table_stride(int pp) {
    const int b = pp * (int)sizeof(T);
    return b <= 64 ? 64 / (int)sizeof(T) : (b <= 128 ? 128 / (int)sizeof(T) : pp);
}
template <typename T>
int launch_grouped_pred_by_id_table(pds_ctx* ctx, const T* const* d_cols, int n_feat, int bias, int64_t n_rows, const int64_t* d_keys,
                                    const int64_t* d_kmin, const uint32_t* d_ids /* dense id of group g */, int64_t n_groups, const T* d_coeffs,
                                    const uint8_t* d_flags, T* d_table /* n_ids x grouped_pred_table_stride(p') workspace */, T* d_pred, T* d_resid, uint8_t* d_row_null);  // grouped_pred.hip
// ---- leverage_mid.hip: HC2 / HC3 leverages of 17 .. 64 f64 features on the matrix cores (PDS_ERR_UNSUPPORTED: not applicable, nothing done)
int launch_grouped_moments_stream(pds_ctx* ctx, const DeviceCols<double>& dc, int n_feat, int64_t n_frame, const int64_t* d_off, int64_t n_groups,
                                  double* d_records);  // grouped_mid.hip: grouped Gram records, 17 .. 64 f64 features, one stream
// grouped_mid.hip: grouped OLS / ridge with 17 .. 32 features as one stream, the solves in the streaming waves (no records)
template <typename T>
int launch_grouped_mid_fused(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_frame, const int64_t* d_off, int64_t n_groups,
                             const SolveParams& sp, T* d_coeffs, uint8_t* d_flags, void* d_ws);
size_t grouped_mid_fused_workspace(int num_cus, int n_feat, int add_bias);
int leverage_operand(pds_ctx* ctx, const double* d_inv, int n_feat, int bias, const double** d_lop);
int launch_report_mid(pds_ctx* ctx, const DeviceCols<double>& dc, int n_feat, int bias, int64_t n_rows, const double* d_beta, const double* d_inv,
                      int hc_mode, double* d_sums, double* d_meat);  // moments_mid.hip: residuals + leverages + meat in one stream
int launch_leverage_mid(pds_ctx* ctx, const DeviceCols<double>& dc, int n_feat, int bias, int64_t n_rows, const double* d_inv, int hc_mode,
                        double* d_s_rows);

// ---- pass2.hip ----
// sum y and sum y^2 of one device column in f64 (fixed-order two-stage reduction): d_out[0..1]
template <typename T>
int launch_y_sums(pds_ctx* ctx, const T* d_y, int64_t n_rows, double* d_out, bool shifted = false /* sums of y - y[0] */);
// streaming residual pass: pred/resid (nullable outputs), sums: [0]=sum e^2, [1]=sum w e^2,
// meat (p' x p') = sum s_i x_i x_i' with s_i = e_i^2 * hc_scale(h_ii) when meat != null.
template <typename T>
int launch_pass2(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, int add_bias,
                 bool weighted, const T* d_beta, const T* d_inv /*nullable, for HC2/3*/, int hc_mode,
                 T* d_pred, T* d_resid, double* d_sums /*[2]*/, double* d_meat /*p'*p' or null*/,
                 // p <= 16: also sum (y - y[0]), sum (y - y[0])^2 -> d_ysums[0..1] (the report's derived var(y), from the rows this pass reads anyway)
                 double* d_ysums = nullptr);

// ---- nulls.hip ----
template <typename T>
struct NullPrepared {
    std::vector<const T*> cols;  // device pointers, reference order [y, x1..xp]
    int64_t n_kept = 0;
    bool dropped = false;         // rows were removed -> d_keep / d_rank are valid
    uint8_t* d_keep = nullptr;
    int64_t* d_rank = nullptr;
};
template <typename T>
int apply_null_policy(pds_ctx* ctx, const std::vector<const T*>& cols_dev, const std::vector<const uint8_t*>& bm_dev,
                      const std::vector<int64_t>& bit_off, int64_t n_rows, int policy, T fill_value,
                      NullPrepared<T>& out);
template <typename T>
int expand_rows(pds_ctx* ctx, const T* d_compact, const uint8_t* d_keep, const int64_t* d_rank, int64_t n_rows, T* d_out,
                uint8_t* d_valid);
size_t null_policy_workspace(int n_cols, int64_t n_rows, size_t elem);
int remap_group_offsets(pds_ctx* ctx, const int64_t* d_off, int64_t n_groups, const int64_t* d_rank, int64_t n_rows,
                        int64_t n_kept, int64_t* d_out);

// ---- rolling.hip ----
template <typename T>
int launch_rolling(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, int add_bias,
                   int64_t window, int64_t min_size, double lambda, bool expanding, const double* seed_moments,
                   T* d_coeffs, T* d_pred, uint8_t* d_valid);

// rolling_wide.hip: 13 .. 64 coefficients (per-row moment records + launch_solve); workspace comes out of ctx->ws
template <typename T>
int launch_rolling_wide(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, int add_bias, int64_t window,
                        int64_t min_size, double lambda, bool expanding, const double* seed_moments, T* d_coeffs, T* d_pred,
                        uint8_t* d_valid);
size_t rolling_wide_workspace(int n_feat, int64_t n_rows, size_t elem);

// ---- keyed.hip: int64 keys in any row order -> sorted keys, permutation, distinct keys, group offsets
int keys_nondecreasing(pds_ctx* ctx, const int64_t* d_keys, int64_t n, unsigned* d_flag, bool* sorted);
size_t keyed_temp_bytes(int64_t n);
size_t keyed_ordered_temp_bytes(int64_t n);
constexpr int kKeySlots = 8192;  // slots of the histogram the order check can take along (keyed.hip / keyed_partition.hip)
int keys_order_minmax(pds_ctx* ctx, const int64_t* d_keys, int64_t n, int64_t* d_state /* 8 slots */, bool* sorted, int64_t* mm,
                      uint32_t* d_run_counts = nullptr /* key_run_slots(n) entries: see keyed_runs_ordered */,
                      unsigned long long* d_run_masks = nullptr /* key_run_mask_bytes(n) */, int64_t* n_runs = nullptr, int hist_shift = -1,
                      unsigned* d_slot_counts = nullptr /* kKeySlots x 8 */, bool* hist_taken = nullptr);
size_t key_run_slots(int64_t n);
size_t key_run_mask_bytes(int64_t n);
int keyed_runs_ordered(pds_ctx* ctx, const int64_t* d_keys, int64_t n, uint32_t* d_counts, uint32_t* d_prefix, const unsigned long long* d_masks,
                       int64_t cap, int64_t* d_unique, int64_t* d_offsets, void* d_temp, size_t temp_bytes);
int keyed_minmax(pds_ctx* ctx, const int64_t* d_keys, int64_t n, void* d_temp, size_t temp_bytes, int64_t* d_minmax, int64_t* mm);
int keyed_sort(pds_ctx* ctx, const int64_t* d_keys, int64_t n, uint32_t* d_idx_in, int64_t* d_sorted_keys, uint32_t* d_perm,
               void* d_temp, size_t temp_bytes, int64_t* d_scratch_keys, const int64_t* d_minmax, const int64_t* mm);
// ---- keyed_partition.hip: grouped moments of a frame in any row order without sorting its rows (dense integer keys)
struct KeyedPartitionState {
    double* table = nullptr;   // [ids][nvp]: upper triangle of Z'Z, Z = [x_0..x_{pc-1}, 1, y], per dense id
    unsigned* ids = nullptr;   // dense id of group r (ascending)
    unsigned* rank = nullptr;  // group of dense id i (valid where the id has rows)
    int pc = 0, nvp = 0;
};
template <typename T>
int64_t keyed_partition_buckets(int n_feat, int64_t n_rows, uint64_t range);  // 0: not applicable (the sorting route)
template <typename T>
size_t keyed_partition_workspace(int n_feat, int64_t n_rows, int64_t n_buckets);
template <typename T>
int keyed_partition_shift(int n_feat);  // log2 of the bucket width (ids per bucket)
template <typename T>
int keyed_partition_build(pds_ctx* ctx, const T* const* d_cols, const int64_t* d_keys, const int64_t* d_kmin /* base: see keyed_partition.hip */,
                          uint64_t range, int n_feat, int64_t n_rows, int64_t n_buckets, char* ws, int64_t max_groups, int64_t* d_out_keys,
                          int64_t* d_offsets, int64_t* n_groups, KeyedPartitionState& st, const unsigned* d_slot_counts = nullptr,
                          unsigned first_slot = 0, int phases = 3 /* 1: the moment table of these rows; 2: the group list from the table */,
                          int64_t n_rows_total = 0 /* rows behind the table when it sums several slices */);
int keyed_partition_add_table(pds_ctx* ctx, const KeyedPartitionState& st, int64_t n_ids, const double* d_other);  // st.table += other
template <typename T>
int64_t keyed_partition_table_ids(int n_feat, int64_t n_buckets);  // rows of the id-indexed table (st.nvp doubles each)
template <typename T>
int keyed_partition_records(pds_ctx* ctx, const KeyedPartitionState& st, int n_feat, int64_t g0, int64_t gc, T* d_records);
int keyed_runs(pds_ctx* ctx, const int64_t* d_sorted_keys, int64_t n, int64_t* d_unique, int64_t* d_counts, int64_t* d_offsets,
               int64_t* d_nruns, void* d_temp, size_t temp_bytes, int64_t* n_groups);
template <typename T>
int launch_gather_rows(pds_ctx* ctx, const T* d_src, const uint32_t* d_perm, int64_t n, T* d_dst);
// layout.hip: rows [0, n_rows) of a row-major device matrix (row stride ld) -> out[c * col_stride + out_row0 + r]
template <typename T>
int launch_rows_to_cols(pds_ctx* ctx, const T* d_X, int64_t ld, int64_t n_rows, int n_cols, T* d_out, int64_t col_stride, int64_t out_row0);
// keyed.hip: the whole frame through the permutation by way of row-major records (one random access per row, not per element)
template <typename T>
int launch_gather_frame(pds_ctx* ctx, const T* const* d_src, const uint32_t* d_perm, int nc, int64_t n, T* d_records, T* const* d_dst);
// the transposition tile of launch_gather_frame (256 rows x nc columns) must fit the 64 KB of LDS a launch gets without raising
// the kernel's dynamic limit; wider frames take the column-by-column gather (launch_gather_rows), which has no width limit
constexpr size_t kGatherFrameLdsLimit = 64 * 1024;
template <typename T>
inline bool gather_frame_fits(int nc) { return (size_t)256 * (size_t)(nc | 1) * sizeof(T) <= kGatherFrameLdsLimit; }
template <typename T>
int launch_scale_sqrt_w(pds_ctx* ctx, const T* d_src /*nullable: ones*/, const T* d_w, int64_t n, T* d_dst);

// ---- stats.cpp ----
double student_t_sf(double x, double df, bool* err);
double student_t_ppf(double q, double df);

}  // namespace pds
