// rolling_wide.hip -- pl_rolling_lr / pl_recursive_lr (/root/reference/src/num_ext/linear_regression.rs:1121-1283,
// drivers lr_online_solvers.rs:148-301) for 13 .. 64 coefficients, where the per-row normal equations no longer fit
// the registers of one lane (rolling.hip: lane = row, p' <= 12).
//
// Same mathematics as rolling.hip -- the window normal equations are rebuilt, never the inverse dragged along --
// in three kernels per chunk of rows:
//   1. rolling_wide_kernel: one wavefront per segment of kSegRows consecutive rows keeps the running augmented
//      moment matrix A = Z'Z (Z = [x | 1 | y], the pds_moments layout) spread over its lanes (entry e = lane + 64 k),
//      adds the row entering the window, subtracts the row leaving it (rows come through a 64-row LDS tile, loaded
//      lane = row, coalesced) and writes one moment record per row.  A segment is anchored exactly: rolling windows
//      are rebuilt from the w rows in front of the segment, expanding ones start from an exclusive prefix over the
//      per-segment totals (first pass + seg_prefix_kernel), so round-off never spans more than kSegRows updates.
//   2. the batched solver of the grouped path (launch_solve: pivoted QR -- the reference's initial fit,
//      faer_qr_lr_with_inv lr_online_solvers.rs:120-143 -- with lambda on every diagonal) on the records.
//   3. rolling_wide_finish_kernel: pred_i = x_i . beta_i, the validity rule (first window-1 rows; the skipping
//      variant's min_size on the finite-row count, which is the [1,1] entry of the record), NaN fill.
// The records of a chunk (q^2 values per row) are sized to stay in the 256 MiB Infinity Cache between 1 and 2.
// Coverage path: correct and parallel, not tuned (p' <= 12 is the measured one).
#include "common.hpp"

namespace pds {

constexpr int kSegRows = 256;
constexpr int kZStride = 65;  // doubles per column of the LDS row tile

struct RollWideArgs {
    int p, pp, bias, q;
    int64_t n, window, min_size;
    int mode;  // 0 rolling, 1 expanding: per-segment totals, 2 expanding: main pass from the prefix
};

template <typename T, int KMAX>
__global__ __launch_bounds__(64) void rolling_wide_kernel(const T* const* __restrict__ cols, RollWideArgs ra, int64_t c0,
                                                          int64_t c1, double* __restrict__ seg_tot,
                                                          T* __restrict__ mom) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x;
    const int p = ra.p, q = ra.q, qq = q * q;
    double* Zn = sm;
    double* Zo = sm + q * kZStride;
    const int64_t s = c0 / kSegRows + blockIdx.x;  // global segment index
    const int64_t r0 = s * kSegRows;
    if (r0 >= c1) return;
    const int64_t r1 = (r0 + kSegRows < c1) ? r0 + kSegRows : c1;
    const int64_t w = ra.window;

    int ia[KMAX], ja[KMAX];  // LDS offsets of the two factors of entry e = lane + 64 k
    double W[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int e = lane + 64 * k;
        const int ee = e < qq ? e : 0;
        ia[k] = (ee % q) * kZStride;
        ja[k] = (ee / q) * kZStride;
        W[k] = (ra.mode == 2 && e < qq) ? seg_tot[s * qq + e] : 0.0;
    }
    int64_t r_begin = r0;
    if (ra.mode == 0) r_begin = r0 - ((w + 63) / 64) * 64;  // warm-up steps rebuild the window in front of the segment
    for (int64_t base = r_begin; base < r1; base += 64) {
        const bool warm = base < r0;
        const int64_t r = base + lane;
        // ---- the 64 rows of this step (and the 64 rows leaving the window), lane = row
        bool fin_n = r >= 0 && r < ra.n && (warm ? r >= r0 - w : r < r1);
        {
            const bool in = fin_n;
            for (int c = 0; c <= p; ++c) {
                const double v = in ? (double)as_global(cols[c])[r] : 0.0;
                fin_n = fin_n && isfinite(v);
                Zn[(c < p ? c : p + 1) * kZStride + lane] = v;
            }
            Zn[p * kZStride + lane] = 1.0;
        }
        bool fin_o = false;
        if (ra.mode == 0 && !warm) {
            const int64_t ro = r - w;
            fin_o = ro >= 0 && r < r1;
            const bool in = fin_o;
            for (int c = 0; c <= p; ++c) {
                const double v = in ? (double)as_global(cols[c])[ro] : 0.0;
                fin_o = fin_o && isfinite(v);
                Zo[(c < p ? c : p + 1) * kZStride + lane] = v;
            }
            Zo[p * kZStride + lane] = 1.0;
        }
        // a row holding a non-finite value is left out of the sums (OnlineLR::update lr_online_solvers.rs:85-89)
        const unsigned long long mask_n = __ballot(fin_n), mask_o = __ballot(fin_o);
        PDS_WAVE_LDS_SYNC();
        const int steps = (int)((r1 - base < 64) ? r1 - base : 64);
        for (int t = 0; t < steps; ++t) {
            if ((mask_n >> t) & 1ull) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) W[k] = fma(Zn[ia[k] + t], Zn[ja[k] + t], W[k]);
            }
            if ((mask_o >> t) & 1ull) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) W[k] = fma(-Zo[ia[k] + t], Zo[ja[k] + t], W[k]);
            }
            if (!warm && ra.mode != 1) {
                T* rec = mom + (base + t - c0) * (int64_t)qq;
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (lane + 64 * k < qq) rec[lane + 64 * k] = (T)W[k];
            }
        }
        PDS_WAVE_LDS_SYNC();
    }
    if (ra.mode == 1) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (lane + 64 * k < qq) seg_tot[s * qq + lane + 64 * k] = W[k];
    }
}

// exclusive prefix over the per-segment totals, thread = matrix entry (fixed order: results do not depend on scheduling)
__global__ __launch_bounds__(256) void seg_prefix_kernel(double* __restrict__ tot, int64_t nseg, int qq,
                                                         const double* __restrict__ seed) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= qq) return;
    double run = seed ? seed[e] : 0.0;
    for (int64_t s = 0; s < nseg; ++s) {
        const double x = tot[s * qq + e];
        tot[s * qq + e] = run;
        run += x;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void rolling_wide_finish_kernel(const T* const* __restrict__ cols, RollWideArgs ra, int64_t c0,
                                                                  int64_t c1, const T* __restrict__ mom,
                                                                  const uint8_t* __restrict__ flags, T* __restrict__ coeffs,
                                                                  T* __restrict__ pred, uint8_t* __restrict__ valid) {
    const int64_t r = c0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= c1) return;
    const int p = ra.p, pp = ra.pp, q = ra.q;
    const double cnt = (double)mom[(r - c0) * (int64_t)(q * q) + p + p * q];  // finite rows in the window
    bool v_ok = r >= ra.window - 1;
    if (ra.min_size > 0) v_ok = v_ok && (cnt >= (double)ra.min_size);
    const bool ok = v_ok && !flags[r - c0];
    T* b = coeffs + r * (int64_t)pp;
    double pr = ra.bias ? (double)b[p] : 0.0;
    for (int a = 0; a < p; ++a) pr = fma((double)as_global(cols[a])[r], (double)b[a], pr);
    const T nanv = (T)__builtin_nan("");
    if (!ok)
        for (int a = 0; a < pp; ++a) b[a] = nanv;
    pred[r] = ok ? (T)pr : nanv;
    valid[r] = v_ok ? 1 : 0;
}

static int64_t wide_chunk_rows(int q, int64_t n_rows, size_t elem) {
    int64_t c = (int64_t)((192ull << 20) / ((size_t)q * q * elem));
    c = std::max<int64_t>(kSegRows, (c / kSegRows) * kSegRows);
    const int64_t n_up = ((n_rows + kSegRows - 1) / kSegRows) * kSegRows;
    return std::min(c, n_up);
}

size_t rolling_wide_workspace(int n_feat, int64_t n_rows, size_t elem) {
    const int q = n_feat + 2;
    const int64_t c = wide_chunk_rows(q, n_rows, elem);
    const int64_t nseg = (n_rows + kSegRows - 1) / kSegRows;
    return (size_t)c * q * q * elem + (size_t)c + (size_t)nseg * q * q * sizeof(double) + (size_t)q * q * sizeof(double) + 8192;
}

template <typename T, int KMAX>
static int launch_wide_k(pds_ctx* ctx, const DeviceCols<T>& dc, RollWideArgs ra, double lambda, bool expanding,
                         const double* seed_moments, T* d_coeffs, T* d_pred, uint8_t* d_valid) {
    const int q = ra.q, qq = q * q;
    const int64_t chunk = wide_chunk_rows(q, ra.n, sizeof(T));
    const int64_t nseg = (ra.n + kSegRows - 1) / kSegRows;
    T* d_mom = reinterpret_cast<T*>(ws_take(ctx, (size_t)chunk * qq * sizeof(T)));
    uint8_t* d_flags = reinterpret_cast<uint8_t*>(ws_take(ctx, (size_t)chunk));
    const size_t lds = (size_t)2 * q * kZStride * sizeof(double);
    auto kern = &rolling_wide_kernel<T, KMAX>;
    if (lds > 64 * 1024)
        PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double* d_tot = nullptr;
    if (expanding) {
        d_tot = reinterpret_cast<double*>(ws_take(ctx, (size_t)nseg * qq * sizeof(double)));
        double* d_seed = nullptr;
        if (seed_moments) {  // rows in front of this frame (row-sharded expanding fit): already in the record layout
            d_seed = reinterpret_cast<double*>(ws_take(ctx, (size_t)qq * sizeof(double)));
            PDS_HIP_CHECK(hipMemcpyAsync(d_seed, seed_moments, (size_t)qq * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
        KernelTimer timer(ctx, kKindRolling);
        ra.mode = 1;
        hipLaunchKernelGGL(kern, dim3((unsigned)nseg), dim3(64), lds, ctx->stream, dc.d_ptrs, ra, (int64_t)0, ra.n, d_tot,
                           (T*)nullptr);
        hipLaunchKernelGGL(seg_prefix_kernel, dim3((qq + 255) / 256), dim3(256), 0, ctx->stream, d_tot, nseg, qq, d_seed);
        PDS_HIP_CHECK(hipGetLastError());
    }
    ra.mode = expanding ? 2 : 0;
    SolveParams sp{ra.p, ra.bias, PDS_SOLVER_QR, lambda, 0.0, 1};
    for (int64_t c0 = 0; c0 < ra.n; c0 += chunk) {
        const int64_t c1 = std::min(ra.n, c0 + chunk);
        {
            KernelTimer timer(ctx, kKindRolling);
            hipLaunchKernelGGL(kern, dim3((unsigned)((c1 - c0 + kSegRows - 1) / kSegRows)), dim3(64), lds, ctx->stream, dc.d_ptrs,
                               ra, c0, c1, d_tot, d_mom);
            PDS_HIP_CHECK(hipGetLastError());
        }
        if (int rc = launch_solve<T>(ctx, d_mom, c1 - c0, sp, d_coeffs + c0 * ra.pp, d_flags, nullptr, nullptr)) return rc;
        KernelTimer timer(ctx, kKindRolling);
        hipLaunchKernelGGL((rolling_wide_finish_kernel<T>), dim3((unsigned)((c1 - c0 + 255) / 256)), dim3(256), 0, ctx->stream,
                           dc.d_ptrs, ra, c0, c1, d_mom, d_flags, d_coeffs, d_pred, d_valid);
        PDS_HIP_CHECK(hipGetLastError());
    }
    return PDS_OK;
}

template <typename T>
int launch_rolling_wide(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, int add_bias, int64_t window,
                        int64_t min_size, double lambda, bool expanding, const double* seed_moments, T* d_coeffs, T* d_pred,
                        uint8_t* d_valid) {
    RollWideArgs ra;
    ra.p = n_feat;
    ra.bias = add_bias ? 1 : 0;
    ra.pp = n_feat + ra.bias;
    ra.q = n_feat + 2;
    ra.n = n_rows;
    ra.window = window;
    ra.min_size = min_size;
    ra.mode = 0;
    const double lam = lambda > 0.0 ? lambda : 0.0;
    const int qq = ra.q * ra.q;
    if (ra.pp > 64 || n_feat > 64)
        return fail(PDS_ERR_UNSUPPORTED, "rolling / recursive: at most 64 coefficients (features + bias) in this build");
    if (qq <= 64 * 6) return launch_wide_k<T, 6>(ctx, dc, ra, lam, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    if (qq <= 64 * 19) return launch_wide_k<T, 19>(ctx, dc, ra, lam, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    return launch_wide_k<T, 69>(ctx, dc, ra, lam, expanding, seed_moments, d_coeffs, d_pred, d_valid);
}

template int launch_rolling_wide<double>(pds_ctx*, const DeviceCols<double>&, int, int64_t, int, int64_t, int64_t, double, bool,
                                         const double*, double*, double*, uint8_t*);
template int launch_rolling_wide<float>(pds_ctx*, const DeviceCols<float>&, int, int64_t, int, int64_t, int64_t, double, bool,
                                        const double*, float*, float*, uint8_t*);

}  // namespace pds
