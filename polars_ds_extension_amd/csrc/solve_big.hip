// solve_big.hip -- normal equations with more than 64 coefficients (the LDS kernels of solve.hip stop there).
//
// `faer_solve_lr` / `faer_solve_lr_gated` (/root/reference/src/linear/lr/lr_solvers.rs:282-382) on a p' x p' Gram
// matrix that no longer fits one wave's LDS: at p' = 512 it is 2 MiB of f64, comfortably L2 resident.  One workgroup
// (1024 threads) per system runs a left-looking column Cholesky on a workspace copy in HBM/L2:
//     column j:  c_i = G_ij - sum_{k<j} L_ik L_jk   (i >= j; thread = row i, the L_ik stream is coalesced down
//                column k, row j of L sits in LDS),  L_jj = sqrt(c_j),  L_ij = c_i / L_jj
// L is kept twice -- column-major (coalesced column reads of the factor loop and of the forward sweep) and row-major
// (coalesced row reads for staging row j and for the backward sweep).  The right-hand side X'y rides along as row p'
// of the matrix, so L z = X'y is solved by the factorisation itself; L' beta = z is an axpy sweep over the rows.
// All arithmetic is f64 whatever the frame's dtype (the reference's f32 path factors in f32).
//
// Solver choice: every `solver=` string lands here on Cholesky.  For the systems the rank gate lets through the
// reference's QR / SVD / LLT agree to rounding x cond; the gate itself is the reference's log-space rule
// (sum ln L_kk^2 - sum ln G_kk <= ln tol  =>  null, non-positive pivot => null, lr_solvers.rs:341-380).  What is NOT
// reproduced beyond 64 coefficients: the finite answers pivoted QR gives for rank-deficient systems with the gate
// switched off -- those come back null here.
//
// The inverse (lin_reg_report's (X'X)^-1, linear_regression.rs:874-880) is p' independent solves against the same
// factor: one workgroup per unit vector, spread over the whole chip.
#include "common.hpp"

namespace pds {

constexpr int kBigThreads = 1024;
constexpr int kInvThreads = 256;

struct BigDev {
    int p, pp, bias, lambda_on_bias, gate_on;
    double lambda, ln_tol;
};

__device__ __forceinline__ double block_sum_1024(double v, double* red /*[16]*/) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];  // fixed order
    return s;
}

// Lc: column-major, ld = pp + 1 rows (row pp = rhs), pp columns.  Lr: row-major, pp + 1 rows of pp entries.
template <typename T>
__global__ __launch_bounds__(kBigThreads) void chol_big_factor_kernel(const T* __restrict__ moments, BigDev sp,
                                                                      double* __restrict__ work, T* __restrict__ coeffs,
                                                                      uint8_t* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int p = sp.p, pp = sp.pp, q = p + 2, ldc = pp + 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t sys = blockIdx.x;
    const T* M = moments + sys * (int64_t)q * q;
    double* Lc = work + sys * (int64_t)2 * ldc * pp;
    double* Lr = Lc + (int64_t)ldc * pp;
    double* rowj = sm;        // pp: row j of L (k < j), later z / beta
    double* red = sm + pp;    // 16
    double* bc = red + 16;    // 2: broadcast slots

    // moment index of coefficient a: features first, then the bias row (index p of the moment matrix)
    auto mi = [&](int a) { return a < p ? a : p; };
    // ---- gate denominator: sum ln G_kk (with lambda), non-positive / non-finite diagonal => null
    double part = 0.0, badp = 0.0;
    for (int k = tid; k < pp; k += nt) {
        double d = (double)M[mi(k) + (int64_t)mi(k) * q];
        if (sp.lambda > 0.0 && (k < p || sp.lambda_on_bias)) d += sp.lambda;
        if (!(d > 0.0) || !isfinite(d)) badp += 1.0;
        else part += log(d);
    }
    const double ln_den = block_sum_1024(part, red);
    const double nbad = block_sum_1024(badp, red);
    bool is_null = sp.gate_on && nbad > 0.0;

    double ln_det = 0.0;
    bool ok = true;
    for (int j = 0; j < pp; ++j) {
        // stage row j of L (columns k < j) in LDS
        for (int k = tid; k < j; k += nt) rowj[k] = Lr[(int64_t)j * pp + k];
        __syncthreads();
        // c_i for rows i = j .. pp (row pp is the right-hand side)
        double cj_mine = 0.0;
        for (int i = j + tid; i <= pp; i += nt) {
            double c;
            if (i < pp) {
                c = (double)M[mi(i) + (int64_t)mi(j) * q];
                if (i == j && sp.lambda > 0.0 && (j < p || sp.lambda_on_bias)) c += sp.lambda;
            } else {
                c = (double)M[mi(j) + (int64_t)(p + 1) * q];  // (X'y)_j
            }
            const double* col = Lc + i;
            // eight independent L2 loads in flight per thread (the loop is latency bound: one workgroup, one system)
            double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            int k = 0;
            for (; k + 8 <= j; k += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = col[(int64_t)(k + u) * ldc];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = fma(v[u], rowj[k + u], acc[u]);
            }
            for (; k < j; ++k) acc[0] = fma(col[(int64_t)k * ldc], rowj[k], acc[0]);
            c -= ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
            Lc[i + (int64_t)j * ldc] = c;  // unscaled for now
            if (i == j) cj_mine = c;
        }
        if (tid == 0) bc[0] = cj_mine;  // i == j belongs to thread 0
        __syncthreads();
        const double d = bc[0];
        if (!(d > 0.0) || !isfinite(d)) {
            ok = false;
            break;  // uniform
        }
        ln_det += log(d);
        const double ljj = sqrt(d);
        for (int i = j + tid; i <= pp; i += nt) {
            const double v = (i == j) ? ljj : Lc[i + (int64_t)j * ldc] / ljj;
            Lc[i + (int64_t)j * ldc] = v;
            Lr[(int64_t)i * pp + j] = v;
        }
        __syncthreads();
    }
    if (!ok) is_null = true;  // "Not positive-definite -> rank-deficient" (lr_solvers.rs:370-371)
    if (sp.gate_on && !is_null && (ln_det - ln_den <= sp.ln_tol)) is_null = true;
    if (flags && tid == 0) flags[sys] = is_null ? 1 : 0;
    if (is_null) {
        for (int k = tid; k < pp; k += nt) coeffs[sys * (int64_t)pp + k] = (T)__builtin_nan("");
        if (tid == 0) work[(int64_t)gridDim.x * 2 * ldc * pp + sys] = 1.0;  // tells the inverse kernel to write NaN
        return;
    }
    if (tid == 0) work[(int64_t)gridDim.x * 2 * ldc * pp + sys] = 0.0;
    // ---- backward sweep L' beta = z, z = row pp of L:  beta_j = z_j / L_jj, then z_i -= L_ji beta_j for i < j
    double* z = rowj;
    __syncthreads();
    for (int k = tid; k < pp; k += nt) z[k] = Lr[(int64_t)pp * pp + k];
    __syncthreads();
    for (int j = pp - 1; j >= 0; --j) {
        const double bj = z[j] / Lr[(int64_t)j * pp + j];
        __syncthreads();
        if (tid == 0) z[j] = bj;
        const double* row = Lr + (int64_t)j * pp;
        for (int i = tid; i < j; i += nt) z[i] = fma(-row[i], bj, z[i]);
        __syncthreads();
    }
    for (int k = tid; k < pp; k += nt) coeffs[sys * (int64_t)pp + k] = (T)z[k];
}

// (G + lambda)^-1 column c: L z = e_c (forward, z_i = 0 above c), L' x = z (backward)
template <typename T>
__global__ __launch_bounds__(kInvThreads) void chol_big_inverse_kernel(const double* __restrict__ work, int pp, int n_sys,
                                                                       T* __restrict__ inv_out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int ldc = pp + 1, c = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int64_t sys = blockIdx.y;
    const double* Lc = work + sys * (int64_t)2 * ldc * pp;
    const double* Lr = Lc + (int64_t)ldc * pp;
    T* out = inv_out + sys * (int64_t)pp * pp + (int64_t)c * pp;
    if (work[(int64_t)n_sys * 2 * ldc * pp + sys] != 0.0) {
        for (int i = tid; i < pp; i += nt) out[i] = (T)__builtin_nan("");
        return;
    }
    double* z = sm;
    for (int i = tid; i < pp; i += nt) z[i] = (i == c) ? 1.0 : 0.0;
    __syncthreads();
    for (int k = c; k < pp; ++k) {
        const double zk = z[k] / Lc[k + (int64_t)k * ldc];
        __syncthreads();
        if (tid == 0) z[k] = zk;
        const double* col = Lc + (int64_t)k * ldc;
        for (int i = k + 1 + tid; i < pp; i += nt) z[i] = fma(-col[i], zk, z[i]);
        __syncthreads();
    }
    for (int j = pp - 1; j >= 0; --j) {
        const double xj = z[j] / Lr[(int64_t)j * pp + j];
        __syncthreads();
        if (tid == 0) z[j] = xj;
        const double* row = Lr + (int64_t)j * pp;
        for (int i = tid; i < j; i += nt) z[i] = fma(-row[i], xj, z[i]);
        __syncthreads();
    }
    for (int i = tid; i < pp; i += nt) out[i] = (T)z[i];
}

template <typename T>
int launch_solve_big(pds_ctx* ctx, const T* d_moments, int64_t n_sys, const SolveParams& sp, T* d_coeffs, uint8_t* d_flags,
                     T* d_inv_out) {
    BigDev sd;
    sd.p = sp.p;
    sd.bias = sp.add_bias ? 1 : 0;
    sd.pp = sp.p + sd.bias;
    sd.lambda_on_bias = sp.lambda_on_bias;
    sd.lambda = sp.lambda;
    sd.gate_on = sp.gate_tol > 0.0 ? 1 : 0;
    sd.ln_tol = sd.gate_on ? std::log(sp.gate_tol) : 0.0;
    const int pp = sd.pp;
    const size_t lds = (size_t)(pp + 32) * sizeof(double);
    if (lds > 160 * 1024) return fail(PDS_ERR_UNSUPPORTED, "solve: more than ~20000 coefficients are not supported");
    if (n_sys > 65535) return fail(PDS_ERR_UNSUPPORTED, "solve: too many systems with more than 64 coefficients");
    const size_t per_sys = (size_t)2 * (pp + 1) * pp * sizeof(double);
    if (int rc = ensure_ws(ctx, ctx->solve_ws, per_sys * (size_t)n_sys + (size_t)n_sys * sizeof(double) + 256)) return rc;
    double* work = reinterpret_cast<double*>(ctx->solve_ws.ptr);
    if (lds > 48 * 1024) {
        PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&chol_big_factor_kernel<T>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&chol_big_inverse_kernel<T>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    KernelTimer timer(ctx, kKindSolve);
    hipLaunchKernelGGL((chol_big_factor_kernel<T>), dim3((unsigned)n_sys), dim3(kBigThreads), lds, ctx->stream, d_moments, sd,
                       work, d_coeffs, d_flags);
    if (d_inv_out)
        hipLaunchKernelGGL((chol_big_inverse_kernel<T>), dim3((unsigned)pp, (unsigned)n_sys), dim3(kInvThreads), lds, ctx->stream,
                           work, pp, (int)n_sys, d_inv_out);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

template int launch_solve_big<double>(pds_ctx*, const double*, int64_t, const SolveParams&, double*, uint8_t*, double*);
template int launch_solve_big<float>(pds_ctx*, const float*, int64_t, const SolveParams&, float*, uint8_t*, float*);

}  // namespace pds
