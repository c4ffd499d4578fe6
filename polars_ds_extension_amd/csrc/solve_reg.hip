// solve_reg.hip -- register-resident batched solve for p' <= 16: the hot solver of the grouped path
// (a million 16x16 systems per call).
//
// Same algorithm as solve.hip (Householder QR with column pivoting on G = X'X + lambda, the reference's
// `col_piv_qr` on the normal equations, lr_solvers.rs:282-294, with the log-det rank gate of :329-382)
// but laid out for the CDNA4 register file instead of LDS:
//   * LPS = 16, 8 or 4 lanes own one system, 64/LPS systems per wavefront, no LDS at all;
//   * lane j holds column j of G (LPS doubles, statically indexed -> VGPRs) and a replica of the rhs;
//   * every cross-lane step is a DPP move: row_newbcast:k broadcasts the pivot column, row_ror /
//     row_half_mirror / quad_perm butterflies do the sub-group reductions.  The only LDS-crossbar
//     traffic is the column swap of the pivot (ds_bpermute);
//   * the reflector is applied in the un-normalised form H = I - v v' / (beta (beta - alpha)),
//     v = x - beta e_k: one sqrt and one reciprocal per step instead of a sqrt and two divisions.
#include "solve_reg_dev.hpp"

namespace pds {

// TRI: the systems come as packed upper triangles over [x_0 .. x_{pc-1}, 1, y] in an id-indexed f64 table (keyed_partition.hip) --
// system s is row ids[s] of the table -- instead of (p+2)^2 records: the expansion pass and its 2 x 800 B per group are gone
template <typename T, int LPS, bool CHOL, bool TRI>
__global__ __launch_bounds__(256) void solve_reg_kernel(const T* __restrict__ moments, int64_t n_sys, SolveRegDev sp,
                                                        T* __restrict__ coeffs, uint8_t* __restrict__ flags,
                                                        const int64_t* __restrict__ rows_per_sys, TriSource tri) {
    constexpr int SPW = 64 / LPS;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPS, j = lane % LPS;
    const int p = sp.p, pp = sp.pp, q = p + 2;
    const int64_t waves_total = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (int64_t base = wave_id * SPW; base < n_sys; base += waves_total * SPW) {
        const int64_t sys = base + sub;
        const bool live = sys < n_sys;
        const T* M = TRI ? nullptr : moments + (live ? sys : 0) * (int64_t)(q * q);
        const double* R = TRI ? tri.table + (size_t)tri.ids[live ? sys : 0] * tri.nvp : nullptr;
        // element (i, k) of the (p+2)^2 moment matrix over [x_0 .. x_{p-1}, 1, y]
        auto elem = [&](int i, int k) __attribute__((always_inline)) -> double {
            if constexpr (TRI) {
                int a = i < p ? i : tri.pc + (i - p), b = k < p ? k : tri.pc + (k - p);
                if (a > b) {
                    const int t = a;
                    a = b;
                    b = t;
                }
                const int qp = tri.pc + 2;
                return R[a * qp - (a * (a - 1)) / 2 + (b - a)];
            } else {
                return (double)M[i + k * q];
            }
        };
        const bool colv = j < pp;
        double dj = colv ? elem(j, j) : 1.0;
        const bool lam = sp.lambda > 0.0 && colv && (j < p || sp.lambda_on_bias);
        if (lam) dj += sp.lambda;
        bool is_null = false;
        if (rows_per_sys && live && rows_per_sys[sys + 1] - rows_per_sys[sys] < pp) is_null = true;
        int pj = j;
        double zj = 0.0;
        if constexpr (CHOL) {
            double a[LPS + 1];
#pragma unroll
            for (int i = 0; i < LPS; ++i) {
                a[i] = (colv && i < pp) ? elem(i, j) : 0.0;
                if (lam && i == j) a[i] += sp.lambda;
            }
            a[LPS] = colv ? elem(j, p + 1) : 0.0;
            chol_core<LPS>(a, dj, j, sp, is_null, zj);
        } else {
            double a[LPS], b[LPS];
#pragma unroll
            for (int i = 0; i < LPS; ++i) {
                a[i] = (colv && i < pp) ? elem(i, j) : 0.0;
                if (lam && i == j) a[i] += sp.lambda;
                b[i] = (i < pp) ? elem(i, p + 1) : 0.0;
            }
            solve_core<LPS>(a, b, dj, j, lane, sp, is_null, pj, zj);
        }
        if (live && colv) coeffs[sys * (int64_t)pp + pj] = is_null ? (T)__builtin_nan("") : (T)zj;
        if (live && j == 0 && flags) flags[sys] = is_null ? 1 : 0;
    }
}

template <typename T, int LPS>
static int launch_lps(pds_ctx* ctx, const T* d_moments, int64_t n_sys, const SolveRegDev& sd, bool chol, T* d_coeffs,
                      uint8_t* d_flags, const int64_t* d_rows, const TriSource* tri = nullptr) {
    constexpr int SPW = 64 / LPS;
    const int64_t waves = (n_sys + SPW - 1) / SPW;
    int64_t nb = (waves + 3) / 4;
    nb = std::min<int64_t>(std::max<int64_t>(nb, 1), (int64_t)ctx->num_cus * 16);
    KernelTimer timer(ctx, kKindSolve);
    const TriSource none{};
    if (tri && chol)
        hipLaunchKernelGGL((solve_reg_kernel<T, LPS, true, true>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_moments, n_sys, sd, d_coeffs,
                           d_flags, d_rows, *tri);
    else if (tri)
        hipLaunchKernelGGL((solve_reg_kernel<T, LPS, false, true>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_moments, n_sys, sd, d_coeffs,
                           d_flags, d_rows, *tri);
    else if (chol)
        hipLaunchKernelGGL((solve_reg_kernel<T, LPS, true, false>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_moments,
                           n_sys, sd, d_coeffs, d_flags, d_rows, none);
    else
        hipLaunchKernelGGL((solve_reg_kernel<T, LPS, false, false>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_moments,
                           n_sys, sd, d_coeffs, d_flags, d_rows, none);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

// returns PDS_ERR_UNSUPPORTED (without setting an error) when the caller should use the LDS kernel
template <typename T>
int launch_solve_reg(pds_ctx* ctx, const T* d_moments, int64_t n_sys, const SolveParams& sp, T* d_coeffs,
                     uint8_t* d_flags, const int64_t* d_rows_per_sys, const TriSource* tri) {
    SolveRegDev sd;
    sd.p = sp.p;
    sd.bias = sp.add_bias ? 1 : 0;
    sd.pp = sp.p + sd.bias;
    sd.lambda_on_bias = sp.lambda_on_bias;
    sd.lambda = sp.lambda;
    sd.gate_on = sp.gate_tol > 0.0 ? 1 : 0;
    sd.ln_tol = sd.gate_on ? std::log(sp.gate_tol) : 0.0;
    sd.inv_tol = sd.gate_on ? 1.0 / sp.gate_tol : HUGE_VAL;
    // Cholesky in registers when asked for and gated (an ungated breakdown must fall back to QR: solve.hip)
    const bool chol = sp.solver == PDS_SOLVER_CHOLESKEY && sd.gate_on;
    if (sd.pp <= 4) return launch_lps<T, 4>(ctx, d_moments, n_sys, sd, chol, d_coeffs, d_flags, d_rows_per_sys, tri);
    if (sd.pp <= 8) return launch_lps<T, 8>(ctx, d_moments, n_sys, sd, chol, d_coeffs, d_flags, d_rows_per_sys, tri);
    return launch_lps<T, 16>(ctx, d_moments, n_sys, sd, chol, d_coeffs, d_flags, d_rows_per_sys, tri);
}

template int launch_solve_reg<double>(pds_ctx*, const double*, int64_t, const SolveParams&, double*, uint8_t*,
                                      const int64_t*, const TriSource*);
template int launch_solve_reg<float>(pds_ctx*, const float*, int64_t, const SolveParams&, float*, uint8_t*,
                                     const int64_t*, const TriSource*);

}  // namespace pds
