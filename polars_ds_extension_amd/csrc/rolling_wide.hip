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

// ---- more than 64 coefficients (the reference's drivers have no limit, lr_online_solvers.rs:148-301): the running moment
// matrix no longer fits the registers of one wavefront, so a segment belongs to a 1024-thread workgroup laid out 32 x 32:
// thread (ti, tj) owns the entries (ti + 32 a, tj + 32 b), a, b < QB = ceil(q / 32) -- per row of the window QB + QB LDS
// reads feed QB^2 FMAs -- and the rows come through 16-row LDS tiles.  Records are solved by the big-system path of
// launch_solve (solve_big.hip: one workgroup per system).  Coverage path: every row costs a (p+2)^2 record and a p'^3 / 3
// factorisation, where the reference drags a p' x p' inverse along at O(p'^2) per row.
constexpr int kBigStep = 16;
constexpr int kBigZStride = kBigStep + 1;

template <typename T, int QB>
__global__ __launch_bounds__(1024) void rolling_big_kernel(const T* const* __restrict__ cols, RollWideArgs ra, int64_t c0,
                                                           int64_t c1, double* __restrict__ seg_tot, T* __restrict__ mom) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, ti = tid & 31, tj = tid >> 5;
    const int p = ra.p, q = ra.q, qq = q * q;
    double* Zn = sm;
    double* Zo = sm + q * kBigZStride;
    int* fin = reinterpret_cast<int*>(sm + 2 * q * kBigZStride);  // [0..15] new rows finite, [16..31] old rows finite
    const int64_t s = c0 / kSegRows + blockIdx.x;
    const int64_t r0 = s * kSegRows;
    if (r0 >= c1) return;
    const int64_t r1 = (r0 + kSegRows < c1) ? r0 + kSegRows : c1;
    const int64_t w = ra.window;
    double W[QB][QB];
#pragma unroll
    for (int a = 0; a < QB; ++a)
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            const int i = ti + 32 * a, j = tj + 32 * b;
            W[a][b] = (ra.mode == 2 && i < q && j < q) ? seg_tot[s * qq + i + (int64_t)j * q] : 0.0;
        }
    int64_t r_begin = r0;
    if (ra.mode == 0) r_begin = r0 - ((w + kBigStep - 1) / kBigStep) * kBigStep;
    for (int64_t base = r_begin; base < r1; base += kBigStep) {
        const bool warm = base < r0;
        // ---- stage kBigStep rows (and the rows leaving the window): thread = (column c, row t)
        if (tid < 32) fin[tid] = 1;
        __syncthreads();
        for (int e = tid; e < (p + 1) * kBigStep; e += 1024) {
            const int c = e / kBigStep, t = e % kBigStep;
            const int slot = c < p ? c : p + 1;
            const int64_t r = base + t;
            const bool in = r >= 0 && r < ra.n && (warm ? r >= r0 - w : r < r1);
            const double v = in ? (double)as_global(cols[c])[r] : 0.0;
            Zn[slot * kBigZStride + t] = v;
            if (!in || !isfinite(v)) fin[t] = 0;  // (benign race: every writer stores 0)
            if (ra.mode == 0 && !warm) {
                const int64_t ro = r - w;
                const bool ino = ro >= 0 && r < r1;
                const double vo = ino ? (double)as_global(cols[c])[ro] : 0.0;
                Zo[slot * kBigZStride + t] = vo;
                if (!ino || !isfinite(vo)) fin[16 + t] = 0;
            }
        }
        if (tid < kBigStep) {
            Zn[p * kBigZStride + tid] = 1.0;
            Zo[p * kBigZStride + tid] = 1.0;
        }
        __syncthreads();
        const int steps = (int)((r1 - base < kBigStep) ? r1 - base : kBigStep);
        for (int t = 0; t < steps; ++t) {
            if (fin[t]) {
                double xi[QB], xj[QB];
#pragma unroll
                for (int a = 0; a < QB; ++a) xi[a] = (ti + 32 * a < q) ? Zn[(ti + 32 * a) * kBigZStride + t] : 0.0;
#pragma unroll
                for (int b = 0; b < QB; ++b) xj[b] = (tj + 32 * b < q) ? Zn[(tj + 32 * b) * kBigZStride + t] : 0.0;
#pragma unroll
                for (int a = 0; a < QB; ++a)
#pragma unroll
                    for (int b = 0; b < QB; ++b) W[a][b] = fma(xi[a], xj[b], W[a][b]);
            }
            if (ra.mode == 0 && !warm && fin[16 + t]) {
                double xi[QB], xj[QB];
#pragma unroll
                for (int a = 0; a < QB; ++a) xi[a] = (ti + 32 * a < q) ? Zo[(ti + 32 * a) * kBigZStride + t] : 0.0;
#pragma unroll
                for (int b = 0; b < QB; ++b) xj[b] = (tj + 32 * b < q) ? Zo[(tj + 32 * b) * kBigZStride + t] : 0.0;
#pragma unroll
                for (int a = 0; a < QB; ++a)
#pragma unroll
                    for (int b = 0; b < QB; ++b) W[a][b] = fma(-xi[a], xj[b], W[a][b]);
            }
            if (!warm && ra.mode != 1) {
                T* rec = mom + (base + t - c0) * (int64_t)qq;
#pragma unroll
                for (int a = 0; a < QB; ++a)
#pragma unroll
                    for (int b = 0; b < QB; ++b) {
                        const int i = ti + 32 * a, j = tj + 32 * b;
                        if (i < q && j < q) rec[i + (int64_t)j * q] = (T)W[a][b];
                    }
            }
        }
        __syncthreads();
    }
    if (ra.mode == 1) {
#pragma unroll
        for (int a = 0; a < QB; ++a)
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                const int i = ti + 32 * a, j = tj + 32 * b;
                if (i < q && j < q) seg_tot[s * qq + i + (int64_t)j * q] = W[a][b];
            }
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

// KMAX > 0: the one-wavefront kernel (q <= 66); QB > 0: the 1024-thread kernel
template <typename T, int KMAX, int QB>
static int launch_wide_k(pds_ctx* ctx, const DeviceCols<T>& dc, RollWideArgs ra, double lambda, bool expanding,
                         const double* seed_moments, T* d_coeffs, T* d_pred, uint8_t* d_valid) {
    const int q = ra.q, qq = q * q;
    const int64_t chunk = wide_chunk_rows(q, ra.n, sizeof(T));
    const int64_t nseg = (ra.n + kSegRows - 1) / kSegRows;
    T* d_mom = reinterpret_cast<T*>(ws_take(ctx, (size_t)chunk * qq * sizeof(T)));
    uint8_t* d_flags = reinterpret_cast<uint8_t*>(ws_take(ctx, (size_t)chunk));
    if (!d_mom || !d_flags) return fail(PDS_ERR_HIP, "workspace allocation failed");
    constexpr bool BIG = QB > 0;
    const size_t lds = BIG ? (size_t)2 * q * kBigZStride * sizeof(double) + 256 : (size_t)2 * q * kZStride * sizeof(double);
    const void* kptr;
    if constexpr (BIG) kptr = reinterpret_cast<const void*>(&rolling_big_kernel<T, QB>);
    else kptr = reinterpret_cast<const void*>(&rolling_wide_kernel<T, KMAX>);
    if (lds > 64 * 1024) PDS_HIP_CHECK(hipFuncSetAttribute(kptr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto run = [&](unsigned nblocks, int64_t c0, int64_t c1, double* tot, T* mom) {
        if constexpr (BIG)
            hipLaunchKernelGGL((rolling_big_kernel<T, QB>), dim3(nblocks), dim3(1024), lds, ctx->stream, dc.d_ptrs, ra, c0, c1, tot, mom);
        else
            hipLaunchKernelGGL((rolling_wide_kernel<T, KMAX>), dim3(nblocks), dim3(64), lds, ctx->stream, dc.d_ptrs, ra, c0, c1, tot, mom);
    };
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
        run((unsigned)nseg, (int64_t)0, ra.n, d_tot, (T*)nullptr);
        hipLaunchKernelGGL(seg_prefix_kernel, dim3((qq + 255) / 256), dim3(256), 0, ctx->stream, d_tot, nseg, qq, d_seed);
        PDS_HIP_CHECK(hipGetLastError());
    }
    ra.mode = expanding ? 2 : 0;
    SolveParams sp{ra.p, ra.bias, PDS_SOLVER_QR, lambda, 0.0, 1};
    for (int64_t c0 = 0; c0 < ra.n; c0 += chunk) {
        const int64_t c1 = std::min(ra.n, c0 + chunk);
        {
            KernelTimer timer(ctx, kKindRolling);
            run((unsigned)((c1 - c0 + kSegRows - 1) / kSegRows), c0, c1, d_tot, d_mom);
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
    if (n_feat > 254)
        return fail(PDS_ERR_UNSUPPORTED, "rolling / recursive: at most 254 feature columns in this build");
    if (qq <= 64 * 6) return launch_wide_k<T, 6, 0>(ctx, dc, ra, lam, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    if (qq <= 64 * 19) return launch_wide_k<T, 19, 0>(ctx, dc, ra, lam, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    if (ra.q <= 66) return launch_wide_k<T, 69, 0>(ctx, dc, ra, lam, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    const int qb = (ra.q + 31) / 32;  // 3 .. 8
    switch (qb) {
        case 3: return launch_wide_k<T, 0, 3>(ctx, dc, ra, lam, expanding, seed_moments, d_coeffs, d_pred, d_valid);
        case 4: return launch_wide_k<T, 0, 4>(ctx, dc, ra, lam, expanding, seed_moments, d_coeffs, d_pred, d_valid);
        case 5: return launch_wide_k<T, 0, 5>(ctx, dc, ra, lam, expanding, seed_moments, d_coeffs, d_pred, d_valid);
        case 6: return launch_wide_k<T, 0, 6>(ctx, dc, ra, lam, expanding, seed_moments, d_coeffs, d_pred, d_valid);
        case 7: return launch_wide_k<T, 0, 7>(ctx, dc, ra, lam, expanding, seed_moments, d_coeffs, d_pred, d_valid);
        default: return launch_wide_k<T, 0, 8>(ctx, dc, ra, lam, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    }
}

template int launch_rolling_wide<double>(pds_ctx*, const DeviceCols<double>&, int, int64_t, int, int64_t, int64_t, double, bool,
                                         const double*, double*, double*, uint8_t*);
template int launch_rolling_wide<float>(pds_ctx*, const DeviceCols<float>&, int, int64_t, int, int64_t, int64_t, double, bool,
                                        const double*, float*, float*, uint8_t*);

}  // namespace pds
