// solve_wave.hip -- batched normal-equation solves with 17 .. 64 features: ONE WAVE PER SYSTEM, the matrix in registers.
//
// `df.group_by(key).agg(pds.lin_reg(...))` with more than 16 features (pl_lr's per-group dispatch, linear_regression.rs:447-497, on
// faer_solve_lr_gated, lr_solvers.rs:329-382) went through the LDS solver of solve.hip -- a pivoted Householder QR of one system
// per workgroup, ~90 / 170 us of dependent LDS round trips per system at 32 / 64 features: 75 % of a grouped fit's time at these
// widths (2 ms per 3 500 systems).  This is the solver of the fused p <= 16 kernel (solve_reg_dev.hpp: square-root-free L D L',
// lane j = column j, the rhs as one more row, the rank gate as a product of pivot ratios) stretched over the whole wave:
//   * lane j holds column j of the CENTRED p x p system (the intercept never takes a lane: G_ij - s_i s_j / n with the column sums
//     the moment record carries, b0 = (sum y - s . beta) / n; the gate's denominators stay the uncentred diagonal -- the same
//     accept / reject rule as on the augmented matrix) -- up to 64 features + intercept fit 64 lanes;
//   * step K broadcasts lane K's entries with v_readlane (compile-time lane: no LDS, no DPP row limits) and every lane j > K updates
//     its rows i > K with one FMA each: 3 instructions per (K, i), ~6 000 for 64 features, no memory on the chain;
//   * clear cases are answered here; a system next to the gate (pivot-ratio product beyond sqrt(1 / tol)), gated, or broken down is
//     MARKED (flag 2, appended to a list) and goes through the reference's default factorisation -- the pivoted QR with the
//     log-det gate of solve.hip -- in a second pass over the marked records only.  solver = "choleskey" IS this factorisation.
#include "common.hpp"
#include "solve_reg_dev.hpp"
#include "solve_wave_dev.hpp"

namespace pds {

namespace {

// PPC: compile-time bound of the feature count (a multiple of 8); p <= PPC features, rows / columns beyond p are exact zeros
template <typename T, int PPC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void solve_wave_kernel(const T* __restrict__ moments, int64_t n_sys, SolveRegDev sp, T* __restrict__ coeffs,
                                                        uint8_t* __restrict__ flags, const int64_t* __restrict__ rows_per_sys,
                                                        int32_t* __restrict__ mark_list, unsigned* __restrict__ mark_count) {
    const int j = threadIdx.x;
    const int p = sp.p, q = p + 2, bias = sp.bias, pout = p + bias;
    const bool colv = j < p;
    const double nanv = __builtin_nan("");
    for (int64_t sys = blockIdx.x; sys < n_sys; sys += gridDim.x) {
        const T* M = moments + sys * (int64_t)(q * q);
        // ---- lane j: column j of G (read as row j: the record is symmetric, so the wave's loads are contiguous)
        double a[PPC + 1];
#pragma unroll
        for (int i = 0; i < PPC; ++i) a[i] = (colv && i < p) ? (double)M[j + i * q] : 0.0;
        a[PPC] = colv ? (double)M[j + (p + 1) * q] : 0.0;
        double dj = colv ? (double)M[j + j * q] : 1.0;
        if (sp.lambda > 0.0 && colv) {
            dj += sp.lambda;
#pragma unroll
            for (int i = 0; i < PPC; ++i)
                if (i == j) a[i] += sp.lambda;
        }
        double sj = 0.0, sy = 0.0, nn = 1.0;
        if (bias) {  // centre
            sj = colv ? (double)M[j + p * q] : 0.0;
            nn = (double)M[p + p * q];
            sy = (double)M[p + (p + 1) * q];
            const double m = sj / nn;
#pragma unroll
            for (int i = 0; i < PPC; ++i) {
                const double si = (i < p) ? (double)M[i + p * q] : 0.0;  // (wave-uniform address)
                a[i] = fma(-si, m, a[i]);
            }
            a[PPC] = fma(-sy, m, a[PPC]);
        }
        const bool few = rows_per_sys && rows_per_sys[sys + 1] - rows_per_sys[sys] < pout;  // "#Data < #features"
        // ---- L D L', rank gate, back substitution (solve_wave_dev.hpp)
        double w;
        bool is_null, suspect;
        wave_ldl_solve<PPC>(a, dj, j, p, few, sp, w, is_null, suspect);
        if (colv) coeffs[sys * (int64_t)pout + j] = is_null ? (T)nanv : (T)w;
        if (bias) {
            const double sb = wave_sum64(colv ? sj * w : 0.0);
            if (j == 0) coeffs[sys * (int64_t)pout + p] = is_null ? (T)nanv : (T)((sy - sb) / nn);
        }
        if (j == 0) {
            if (flags) flags[sys] = suspect ? 2 : (is_null ? 1 : 0);
            if (suspect && mark_list) mark_list[atomicAdd(mark_count, 1u)] = (int32_t)sys;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_records_idx_kernel(const T* __restrict__ src, const int32_t* __restrict__ list, int64_t n, int qq,
                                                                 T* __restrict__ dst, const int64_t* __restrict__ rows_src,
                                                                 int64_t* __restrict__ rows_dst /* n + 1 offsets */) {
    const int64_t total = n * qq;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e / qq;
        dst[e] = src[(int64_t)list[k] * qq + (e - k * qq)];
    }
    // row counts of the compacted systems as the offsets the LDS solver takes (n <= kWaveSecondChunk: one thread, in order)
    if (rows_src && blockIdx.x == 0 && threadIdx.x == 0) {
        int64_t acc = 0;
        rows_dst[0] = 0;
        for (int64_t k = 0; k < n; ++k) {
            const int64_t g = list[k];
            acc += rows_src[g + 1] - rows_src[g];
            rows_dst[k + 1] = acc;
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void scatter_results_idx_kernel(const T* __restrict__ co_c, const uint8_t* __restrict__ fl_c,
                                                                  const int32_t* __restrict__ list, int64_t n, int pp, T* __restrict__ coeffs,
                                                                  uint8_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * pp) return;
    const int64_t k = i / pp;
    const int c = (int)(i - k * pp);
    const int64_t g = list[k];
    coeffs[g * pp + c] = co_c[i];
    if (c == 0 && flags) flags[g] = fl_c[k] ? 1 : 0;
}

template <typename T, int PPC>
void launch_ppc(dim3 g, hipStream_t st, const T* mom, int64_t n_sys, const SolveRegDev& sd, T* co, uint8_t* fl, const int64_t* rows, int32_t* list,
                unsigned* count) {
    hipLaunchKernelGGL((solve_wave_kernel<T, PPC>), g, dim3(64), 0, st, mom, n_sys, sd, co, fl, rows, list, count);
}

}  // namespace

constexpr int kWaveSecondChunk = 512;  // marked records per pass of the LDS solver

size_t solve_wave_workspace(int n_feat, int add_bias, int64_t n_sys, size_t elem) {
    const size_t q = (size_t)n_feat + 2, pp = (size_t)n_feat + (add_bias ? 1 : 0);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return 256 + up((size_t)n_sys * 4) + up((size_t)kWaveSecondChunk * q * q * elem) + up((size_t)kWaveSecondChunk * pp * elem) +
           up((size_t)kWaveSecondChunk) + up(((size_t)kWaveSecondChunk + 1) * 8) + 256;
}

// OLS / ridge systems of 17 .. 64 features from (p+2)^2 moment records: PDS_ERR_UNSUPPORTED (nothing done) when this solver does
// not apply (gate off, lambda on the intercept, other widths) -- the caller then keeps launch_solve.
template <typename T>
int launch_solve_wave(pds_ctx* ctx, const T* d_moments, int64_t n_sys, const SolveParams& sp, T* d_coeffs, uint8_t* d_flags,
                      const int64_t* d_rows_per_sys, void* d_ws /* solve_wave_workspace(...) bytes, reusable across calls */) {
    if (sp.p <= 16 || sp.p > 64 || !(sp.gate_tol > 0.0) || sp.lambda_on_bias || !d_flags || !d_ws) return PDS_ERR_UNSUPPORTED;
    static const bool off = [] { const char* e = dev_env("PDS_SOLVE_WAVE"); return e && e[0] == '0'; }();  // (A/B)
    if (off) return PDS_ERR_UNSUPPORTED;
    if (n_sys <= 0) return PDS_OK;
    SolveRegDev sd;
    sd.p = sp.p;
    sd.bias = sp.add_bias ? 1 : 0;
    sd.pp = sp.p;
    sd.lambda_on_bias = 0;
    sd.lambda = sp.lambda;
    sd.gate_on = 1;
    sd.ln_tol = std::log(sp.gate_tol);
    sd.inv_tol = 1.0 / sp.gate_tol;
    const bool second_pass = sp.solver != PDS_SOLVER_CHOLESKEY;
    sd.sus_tol = second_pass ? std::sqrt(sd.inv_tol) : 0.0;
    sd.sus_ratio = second_pass ? solve_suspect_ratio() : 0.0;
    sd.sus_band = 1e-5;
    int32_t* d_list = nullptr;
    unsigned* d_count = nullptr;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    char* wsp = static_cast<char*>(d_ws);
    wsp += (256 - (reinterpret_cast<uintptr_t>(wsp) & 255)) & 255;
    auto take = [&](size_t b) { char* r = wsp; wsp += up(b); return r; };
    if (second_pass) {
        d_count = reinterpret_cast<unsigned*>(take(256));
        d_list = reinterpret_cast<int32_t*>(take((size_t)n_sys * sizeof(int32_t)));
        PDS_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned), ctx->stream));
    }
    const int nb = (int)std::min<int64_t>(n_sys, (int64_t)ctx->num_cus * (sp.p <= 32 ? 16 : 8));  // (up to 32 features: ~100 registers, four waves per SIMD)
    {
        KernelTimer timer(ctx, kKindSolve);
        const dim3 g((unsigned)nb);
        const int p = sp.p;
        if (p <= 24) launch_ppc<T, 24>(g, ctx->stream, d_moments, n_sys, sd, d_coeffs, d_flags, d_rows_per_sys, d_list, d_count);
        else if (p <= 32) launch_ppc<T, 32>(g, ctx->stream, d_moments, n_sys, sd, d_coeffs, d_flags, d_rows_per_sys, d_list, d_count);
        else if (p <= 40) launch_ppc<T, 40>(g, ctx->stream, d_moments, n_sys, sd, d_coeffs, d_flags, d_rows_per_sys, d_list, d_count);
        else if (p <= 48) launch_ppc<T, 48>(g, ctx->stream, d_moments, n_sys, sd, d_coeffs, d_flags, d_rows_per_sys, d_list, d_count);
        else if (p <= 56) launch_ppc<T, 56>(g, ctx->stream, d_moments, n_sys, sd, d_coeffs, d_flags, d_rows_per_sys, d_list, d_count);
        else launch_ppc<T, 64>(g, ctx->stream, d_moments, n_sys, sd, d_coeffs, d_flags, d_rows_per_sys, d_list, d_count);
        PDS_HIP_CHECK(hipGetLastError());
    }
    if (!second_pass) return PDS_OK;
    unsigned h_count = 0;
    PDS_HIP_CHECK(hipMemcpyAsync(&h_count, d_count, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const int64_t marked = (int64_t)h_count;
    if (marked == 0) return PDS_OK;
    const int pp_all = sp.p + (sp.add_bias ? 1 : 0);
    if (marked * 4 > n_sys && pp_all <= 64) {
        // most of the chunk sits next to the gate (e.g. groups with barely more rows than columns): the pivoted QR over the whole
        // chunk in place costs what it always did; compacting it 512 records at a time would cost twice that
        return launch_solve_marked<T>(ctx, d_moments, n_sys, sp, d_coeffs, d_flags, d_rows_per_sys);
    }
    // ---- marked systems: their records, compacted, through the pivoted QR with the log-det gate (solve.hip), results scattered back
    const int q = sp.p + 2, pp = sp.p + (sp.add_bias ? 1 : 0);
    const int64_t chunk = std::min<int64_t>(kWaveSecondChunk, marked);
    T* rec_c = reinterpret_cast<T*>(take((size_t)kWaveSecondChunk * q * q * sizeof(T)));
    T* co_c = reinterpret_cast<T*>(take((size_t)kWaveSecondChunk * pp * sizeof(T)));
    uint8_t* fl_c = reinterpret_cast<uint8_t*>(take((size_t)kWaveSecondChunk));
    int64_t* rows_c = reinterpret_cast<int64_t*>(take(((size_t)kWaveSecondChunk + 1) * 8));
    SolveParams sq = sp;
    sq.solver = PDS_SOLVER_QR;
    for (int64_t k0 = 0; k0 < marked; k0 += chunk) {
        const int64_t kc = std::min(chunk, marked - k0);
        const int gb = (int)std::min<int64_t>((kc * q * q + 255) / 256, (int64_t)ctx->num_cus * 16);
        hipLaunchKernelGGL((gather_records_idx_kernel<T>), dim3(gb), dim3(256), 0, ctx->stream, d_moments, (const int32_t*)(d_list + k0), kc, q * q,
                           rec_c, d_rows_per_sys, rows_c);
        // (64 features + intercept = 65 coefficients is beyond the LDS solver: the big-system solver takes those -- it has no
        //  row-count rule, and needs none: systems with too few rows were answered above and are never marked)
        if (int rc = launch_solve_marked<T>(ctx, rec_c, kc, sp, co_c, fl_c, (d_rows_per_sys && pp <= 64) ? rows_c : nullptr)) return rc;
        hipLaunchKernelGGL((scatter_results_idx_kernel<T>), dim3((unsigned)((kc * pp + 255) / 256)), dim3(256), 0, ctx->stream, (const T*)co_c,
                           (const uint8_t*)fl_c, (const int32_t*)(d_list + k0), kc, pp, d_coeffs, d_flags);
        PDS_HIP_CHECK(hipGetLastError());
    }
    return PDS_OK;
}
template int launch_solve_wave<double>(pds_ctx*, const double*, int64_t, const SolveParams&, double*, uint8_t*, const int64_t*, void*);
template int launch_solve_wave<float>(pds_ctx*, const float*, int64_t, const SolveParams&, float*, uint8_t*, const int64_t*, void*);

}  // namespace pds
