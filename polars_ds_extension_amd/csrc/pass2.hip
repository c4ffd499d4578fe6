// pass2.hip -- the second streaming pass of pl_lr_pred / pl_lin_reg_report / pl_wls_report:
//   pred = X b, resid = y - pred                      (linear_regression.rs:782-785, 863, 1036)
//   sum e^2, sum w e^2                                 (:866-879, 1037-1042)
//   s_i = e_i^2 * {1, 1/(1-h_ii), 1/(1-h_ii)^2}        (HC0/1, HC2, HC3 :880-909), h_ii = x_i' (X'X)^-1 x_i
// The HC "meat" sum_i s_i x_i x_i' is then ONE MORE weighted Gram build (moments.hip with w = s), so the
// reference's p' x N temporary (X'X)^-1 X' (:857) never exists.
//
// Layout: lane = row (RPL consecutive rows per lane, 16-byte loads, 1 KiB coalesced per instruction);
// b and (X'X)^-1 are wave-uniform and come through the scalar cache.  HBM-bound: reads N(p+1)[+1]
// elements, writes 0..3 N.
#include "common.hpp"

#include <type_traits>

namespace pds {

template <typename T>
struct V16;
template <>
struct V16<double> {
    typedef double type __attribute__((ext_vector_type(2), aligned(8)));
    static constexpr int RPL = 2;
};
template <>
struct V16<float> {
    typedef float type __attribute__((ext_vector_type(4), aligned(4)));
    static constexpr int RPL = 4;
};

constexpr int kP2Threads = 256;

// PC: the feature count as a compile-time constant (16, 8, 4, 2, 1; 0 = run time): the per-column `c < p` scalar branches
// in the row loop fold away
#define PDS_P2_BLOCKS 2
template <typename T, bool WEIGHTED, int HC, int PC>
__global__ __launch_bounds__(kP2Threads, PDS_P2_BLOCKS) void pass2_kernel(const T* const* __restrict__ cols, int p_arg, int bias,
                                                           int64_t n, const T* __restrict__ beta,
                                                           const T* __restrict__ inv, T* __restrict__ pred_out,
                                                           T* __restrict__ resid_out, T* __restrict__ s_out,
                                                           double* __restrict__ partials, double* __restrict__ ypart) {
    using V = typename V16<T>::type;
    constexpr int RPL = V16<T>::RPL;
    const int p = PC ? PC : p_arg;
    const int pp = p + bias;
    double sse = 0.0, wsse = 0.0;
    __shared__ T s_inv[HC >= 2 ? 17 * 17 : 1];  // (X'X)^-1 padded to 17 x 17, off-diagonal entries doubled (HC2 / HC3 leverages)
    if constexpr (HC >= 2) {
        for (int i = threadIdx.x; i < 17 * 17; i += blockDim.x) {
            const int a = i % 17, b = i / 17;
            s_inv[i] = (a < pp && b < pp) ? inv[a + b * pp] * (a == b ? T(1) : T(2)) : T(0);
        }
        __syncthreads();
    }
    // loop-invariant, wave-uniform: column pointers and coefficients live in SGPRs
    gptr<T> cx[16];
    T bx[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        cx[c] = as_global(cols[c < p ? c : 0]);
        bx[c] = (c < p) ? beta[c] : T(0);
    }
    const gptr<T> cy = as_global(cols[p]);
    const gptr<T> cw = as_global(WEIGHTED ? cols[p + 1] : cols[p]);
    const T b0 = bias ? beta[p] : T(0);
    // ypart (the report's derived var(y), PDS_REPORT_DERIVE_YVAR): sums of y - y[0] and its square over the rows this pass reads anyway
    const double yc = ypart ? (double)cy[0] : 0.0;
    double sy = 0.0, syy = 0.0;
    // Wave w owns the CONTIGUOUS chunk range [nchunk w / W, nchunk (w+1) / W) (a chunk = 64 RPL rows = 1 KiB of every column):
    // consecutive pieces of a column stay with one wave, the loads are non-temporal (every element is read once) and the next
    // chunk is in flight in a second register set while this one is computed -- the recipe of the Gram kernel (moments.hip),
    // which took this kernel's access pattern from 0.60 to 0.8 of the HBM peak.  The ragged last chunk belongs to the last wave.
    constexpr int CH = 64 * RPL;
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t nfull = n / CH;
    int64_t t = (int64_t)(((__int128)nfull * wid) / nw);
    const int64_t t_end = (int64_t)(((__int128)nfull * (wid + 1)) / nw);
    V xa[16], ya, wa;
    auto load_full = [&](int64_t row, V (&x)[16], V& yv, V& wv) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < p) x[c] = __builtin_nontemporal_load(reinterpret_cast<gptr<V>>(cx[c] + row));
        yv = __builtin_nontemporal_load(reinterpret_cast<gptr<V>>(cy + row));
        if (WEIGHTED) wv = __builtin_nontemporal_load(reinterpret_cast<gptr<V>>(cw + row));
    };
    auto compute = [&](int64_t row, const V (&x)[16], const V& yv, const V& wv, bool full) __attribute__((always_inline)) {
        V pr, rs, sv;
        T om_of[RPL];
        if constexpr (HC >= 2) {
            // leverage h = z' A z, z = [x_0 .. x_{p-1}, 1 (bias)], over the upper triangle of A = (X'X)^-1 (off-diagonal entries stored
            // doubled): every entry is ONE broadcast read from the block's LDS copy, used for all of the lane's rows.  (As scalar loads
            // the compiler kept all 17 x 17 entries in scalar registers: 1 310 spilled SGPRs + scratch in the 16-feature variant.)
            T hh[RPL];
#pragma unroll
            for (int e = 0; e < RPL; ++e) hh[e] = T(0);
            // up to 8 features the 45 entries stay in registers across the chunks (the compiler hoists the reads); beyond, they are read
            // again for every chunk -- the 153 entries of 16 features + intercept do not fit beside two register sets of the frame
            // (hoisted: 390 spilled VGPRs).  An empty asm makes the base a per-chunk value.
            int zo = 0;
            if constexpr (PC == 0 || PC > 8) asm volatile("" : "+v"(zo));
            const T* const sinv = s_inv + zo;
#pragma unroll
            for (int a = 0; a < 17; ++a) {
                if (a < p || (bias && a == p)) {
                    T tt[RPL];
                    const T daa = sinv[a + 17 * a];
#pragma unroll
                    for (int e = 0; e < RPL; ++e) tt[e] = daa * (a < p ? x[a < 16 ? a : 0][e] : T(1));
#pragma unroll
                    for (int b = a + 1; b < 17; ++b)
                        if (b < p || (bias && b == p)) {
                            const T dab = sinv[a + 17 * b];
#pragma unroll
                            for (int e = 0; e < RPL; ++e) tt[e] = fma(dab, (b < p ? x[b < 16 ? b : 0][e] : T(1)), tt[e]);
                        }
#pragma unroll
                    for (int e = 0; e < RPL; ++e) hh[e] = fma(a < p ? x[a < 16 ? a : 0][e] : T(1), tt[e], hh[e]);
                    if constexpr (PC == 0 || PC > 8) __builtin_amdgcn_sched_barrier(0);  // (one row of A at a time: the reads of all 17 rows ahead of their use filled the register file)
                }
            }
#pragma unroll
            for (int e = 0; e < RPL; ++e) om_of[e] = T(1) - hh[e];
        }
#pragma unroll
        for (int e = 0; e < RPL; ++e) {
            T acc = b0;
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < p) acc += x[c][e] * bx[c];
            pr[e] = acc;
            const T r = yv[e] - acc;
            rs[e] = r;
            const bool in = full || row + e < n;
            const double rd = in ? (double)r : 0.0;
            sse = fma(rd, rd, sse);
            if (ypart) {
                const double dy = in ? (double)yv[e] - yc : 0.0;
                sy += dy;
                syy = fma(dy, dy, syy);
            }
            if (WEIGHTED) wsse = fma((double)wv[e], rd * rd, wsse);
            if (HC) {
                T s = r * r;
                if (HC >= 2) s = (HC == 2) ? s * (T(1) / om_of[e]) : s * (T(1) / (om_of[e] * om_of[e]));
                sv[e] = in ? s : T(0);
            }
        }
        if (full) {
            if (pred_out) __builtin_nontemporal_store(pr, reinterpret_cast<V*>(pred_out + row));
            if (resid_out) __builtin_nontemporal_store(rs, reinterpret_cast<V*>(resid_out + row));
            if (HC) *reinterpret_cast<V*>(s_out + row) = sv;  // (re-read by the weighted Gram build right behind this kernel)
        } else {
#pragma unroll
            for (int e = 0; e < RPL; ++e)
                if (row + e < n) {
                    if (pred_out) pred_out[row + e] = pr[e];
                    if (resid_out) resid_out[row + e] = rs[e];
                    if (HC) s_out[row + e] = sv[e];
                }
        }
    };
    // (one register set at four waves per SIMD instead -- 98 VGPRs -- measured 2.12 vs 2.08 ms: the double buffer stays)
    // (the HC2 / HC3 form at 9+ features computes ~350 f64 operations per row: one register set -- with the second one the
    //  16-feature variant spilled 13 registers to scratch; its loads are covered by the other waves of the SIMD)
    constexpr bool DOUBLE_SET = !(HC >= 2 && (PC == 0 || PC > 8));
    if (DOUBLE_SET && t < t_end) load_full(t * CH + lane * RPL, xa, ya, wa);
    for (; t < t_end; ++t) {
        if constexpr (DOUBLE_SET) {
            V xb[16], yb, wb;
#pragma unroll
            for (int c = 0; c < 16; ++c) xb[c] = xa[c];
            yb = ya;
            wb = wa;
            if (t + 1 < t_end) load_full((t + 1) * CH + lane * RPL, xa, ya, wa);
            compute(t * CH + lane * RPL, xb, yb, wb, true);
        } else {
            load_full(t * CH + lane * RPL, xa, ya, wa);
            compute(t * CH + lane * RPL, xa, ya, wa, true);
        }
    }
    if (nfull * CH < n && wid == nw - 1) {  // ragged tail: exactly one wave
        const int64_t row = nfull * CH + lane * RPL;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < p) {
#pragma unroll
                for (int e = 0; e < RPL; ++e) xa[c][e] = (row + e < n) ? cx[c][row + e] : T(0);
            }
#pragma unroll
        for (int e = 0; e < RPL; ++e) ya[e] = (row + e < n) ? cy[row + e] : T(0);
        if (WEIGHTED) {
#pragma unroll
            for (int e = 0; e < RPL; ++e) wa[e] = (row + e < n) ? cw[row + e] : T(0);
        }
        compute(row, xa, ya, wa, false);
    }
    // block reduction (fixed order) -> one partial pair per block
    __shared__ double red[4][kP2Threads / 64];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        sse += __shfl_xor(sse, o);
        wsse += __shfl_xor(wsse, o);
        sy += __shfl_xor(sy, o);
        syy += __shfl_xor(syy, o);
    }
    const int wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = sse;
        red[1][wave] = wsse;
        red[2][wave] = sy;
        red[3][wave] = syy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
        for (int w = 0; w < kP2Threads / 64; ++w) {
            a += red[0][w];
            b += red[1][w];
            c += red[2][w];
            d += red[3][w];
        }
        partials[2 * blockIdx.x] = a;
        partials[2 * blockIdx.x + 1] = b;
        if (ypart) {
            ypart[2 * blockIdx.x] = c;
            ypart[2 * blockIdx.x + 1] = d;
        }
    }
}

__global__ void pass2_finalize_kernel(const double* __restrict__ partials, int nblocks, double* __restrict__ sums) {
    // single wave, fixed order
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 64) {
        a += partials[2 * i];
        b += partials[2 * i + 1];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    if (threadIdx.x == 0) {
        sums[0] = a;
        sums[1] = b;
    }
}

// p > 16: same pass with a run-time loop over the columns (beta from the scalar cache), SE / pred / resid only
template <typename T, bool WEIGHTED>
__global__ __launch_bounds__(kP2Threads) void pass2_wide_kernel(const T* const* __restrict__ cols, int p, int bias,
                                                                int64_t n, const T* __restrict__ beta,
                                                                T* __restrict__ pred_out, T* __restrict__ resid_out,
                                                                T* __restrict__ s_out /* e^2 per row (HC0 / HC1) */,
                                                                double* __restrict__ partials) {
    double sse = 0.0, wsse = 0.0;
    const gptr<T> cy = as_global(cols[p]);
    const gptr<T> cw = as_global(WEIGHTED ? cols[p + 1] : cols[p]);
    const T b0 = bias ? beta[p] : T(0);
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        T acc = b0;
        for (int c = 0; c < p; ++c) acc += as_global(cols[c])[r] * beta[c];
        const T res = cy[r] - acc;
        if (pred_out) pred_out[r] = acc;
        if (resid_out) resid_out[r] = res;
        if (s_out) s_out[r] = res * res;
        const double rd = (double)res;
        sse = fma(rd, rd, sse);
        if (WEIGHTED) wsse = fma((double)cw[r], rd * rd, wsse);
    }
    __shared__ double red[2][kP2Threads / 64];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        sse += __shfl_xor(sse, o);
        wsse += __shfl_xor(wsse, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = sse;
        red[1][wave] = wsse;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < kP2Threads / 64; ++w) {
            a += red[0][w];
            b += red[1][w];
        }
        partials[2 * blockIdx.x] = a;
        partials[2 * blockIdx.x + 1] = b;
    }
}

// p > 16, HC2 / HC3: the leverages h_i = z_i' (X'X)^-1 z_i (z = [x, (1)]; linear_regression.rs:893-909 takes them from
// the diagonal of X (X'X)^-1 X') scale the squared residuals the wide pass left in s: s_i /= (1 - h_i) or (1 - h_i)^2.
// lane = row.  LDS = true: the inverse sits in LDS (f64, rows padded to a multiple of 16) and the row's quadratic form runs in
// chunks of 16 output coordinates -- per b one value of the row from memory and 16 LDS broadcasts feeding 16 independent
// accumulators; every value of the row is read p' / 16 times.  (The first form -- inverse entries through the scalar cache,
// one dependent chain per a, the row re-read p' times -- took 100 ms for 2e7 x 64; it stays for inverses beyond the LDS.)
// ... and for p' <= 80 all ceil(p' / 16) chunks at once: the row is read ONCE (one column pointer + one value per b), every
// value feeds 16 NC independent accumulators (NC <= 5: 160 registers of f64)
template <typename T, int NC>
__global__ __launch_bounds__(kP2Threads) void leverage_scale_wide_regs_kernel(const T* const* __restrict__ cols, int p, int bias,
                                                                              int64_t n, const T* __restrict__ inv, int hc,
                                                                              T* __restrict__ s_rows) {
    extern __shared__ __attribute__((aligned(16))) double Al[];
    const int pp = p + bias;
    constexpr int ps = NC * 16;
    for (int i = threadIdx.x; i < pp * ps; i += blockDim.x) {
        const int b = i / ps, a = i - b * ps;
        Al[i] = a < pp ? (double)inv[a + (size_t)b * pp] : 0.0;
    }
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        double t[ps];
#pragma unroll
        for (int a = 0; a < ps; ++a) t[a] = 0.0;
        double h = 0.0;
        for (int b = 0; b < pp; ++b) {
            const double zb = b < p ? (double)as_global(cols[b])[r] : 1.0;
            const double* Ab = Al + b * ps;
#pragma unroll
            for (int a = 0; a < ps; ++a) t[a] = fma(Ab[a], zb, t[a]);
        }
        // h = z' t: a second trip over the row (L1 / L2 hits), fully unrolled so that t stays in registers
#pragma unroll
        for (int a = 0; a < ps; ++a) {
            if (a < pp) {
                const double za = a < p ? (double)as_global(cols[a])[r] : 1.0;
                h = fma(za, t[a], h);
            }
        }
        const double om = 1.0 - h;
        const double sc = (hc == 2) ? 1.0 / om : 1.0 / (om * om);
        s_rows[r] = (T)((double)s_rows[r] * sc);
    }
}

template <typename T, bool LDS>
__global__ __launch_bounds__(kP2Threads) void leverage_scale_wide_kernel(const T* const* __restrict__ cols, int p, int bias,
                                                                         int64_t n, const T* __restrict__ inv, int hc,
                                                                         T* __restrict__ s_rows) {
    const int pp = p + bias;
    if constexpr (LDS) {
        extern __shared__ __attribute__((aligned(16))) double Al[];
        const int ps = (pp + 15) & ~15;  // padded row: the chunks of 16 never test their tail
        for (int i = threadIdx.x; i < pp * ps; i += blockDim.x) {
            const int b = i / ps, a = i - b * ps;
            Al[i] = a < pp ? (double)inv[a + (size_t)b * pp] : 0.0;
        }
        __syncthreads();
        for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
            double h = 0.0;
            for (int a0 = 0; a0 < pp; a0 += 16) {
                double t[16];
#pragma unroll
                for (int a = 0; a < 16; ++a) t[a] = 0.0;
                for (int b = 0; b < pp; ++b) {
                    const double zb = b < p ? (double)as_global(cols[b])[r] : 1.0;
                    const double* Ab = Al + b * ps + a0;
#pragma unroll
                    for (int a = 0; a < 16; ++a) t[a] = fma(Ab[a], zb, t[a]);
                }
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    if (a0 + a < pp) {
                        const double za = a0 + a < p ? (double)as_global(cols[a0 + a])[r] : 1.0;
                        h = fma(za, t[a], h);
                    }
                }
            }
            const double om = 1.0 - h;
            const double sc = (hc == 2) ? 1.0 / om : 1.0 / (om * om);
            s_rows[r] = (T)((double)s_rows[r] * sc);
        }
    } else {
        for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
            double h = 0.0;
            for (int a = 0; a < pp; ++a) {
                double t = bias ? (double)inv[a + (size_t)p * pp] : 0.0;
                const T* ia = inv + (size_t)a * pp;  // row a = column a (symmetric)
                for (int b = 0; b < p; ++b) t = fma((double)ia[b], (double)as_global(cols[b])[r], t);
                const double za = a < p ? (double)as_global(cols[a])[r] : 1.0;
                h = fma(za, t, h);
            }
            const double om = 1.0 - h;
            const double sc = (hc == 2) ? 1.0 / om : 1.0 / (om * om);
            s_rows[r] = (T)((double)s_rows[r] * sc);
        }
    }
}

template <typename T, bool W, int P16>
static void launch_p2_w(int hc, dim3 g, hipStream_t st, const T* const* cols, int p, int bias, int64_t n, const T* beta,
                        const T* inv, T* pred, T* resid, T* s, double* partials, double* ypart) {
    switch (hc) {
        case 0: hipLaunchKernelGGL((pass2_kernel<T, W, 0, P16>), g, dim3(kP2Threads), 0, st, cols, p, bias, n, beta, inv, pred, resid, s, partials, ypart); break;
        case 1: hipLaunchKernelGGL((pass2_kernel<T, W, 1, P16>), g, dim3(kP2Threads), 0, st, cols, p, bias, n, beta, inv, pred, resid, s, partials, ypart); break;
        case 2: hipLaunchKernelGGL((pass2_kernel<T, W, 2, P16>), g, dim3(kP2Threads), 0, st, cols, p, bias, n, beta, inv, pred, resid, s, partials, ypart); break;
        default: hipLaunchKernelGGL((pass2_kernel<T, W, 3, P16>), g, dim3(kP2Threads), 0, st, cols, p, bias, n, beta, inv, pred, resid, s, partials, ypart); break;
    }
}
template <typename T, bool W>
static void launch_p2(int hc, dim3 g, hipStream_t st, const T* const* cols, int p, int bias, int64_t n, const T* beta,
                      const T* inv, T* pred, T* resid, T* s, double* partials, double* ypart) {
    switch (p) {
        case 16: launch_p2_w<T, W, 16>(hc, g, st, cols, p, bias, n, beta, inv, pred, resid, s, partials, ypart); break;
        case 8: launch_p2_w<T, W, 8>(hc, g, st, cols, p, bias, n, beta, inv, pred, resid, s, partials, ypart); break;
        case 4: launch_p2_w<T, W, 4>(hc, g, st, cols, p, bias, n, beta, inv, pred, resid, s, partials, ypart); break;
        case 2: launch_p2_w<T, W, 2>(hc, g, st, cols, p, bias, n, beta, inv, pred, resid, s, partials, ypart); break;
        case 1: launch_p2_w<T, W, 1>(hc, g, st, cols, p, bias, n, beta, inv, pred, resid, s, partials, ypart); break;
        default: launch_p2_w<T, W, 0>(hc, g, st, cols, p, bias, n, beta, inv, pred, resid, s, partials, ypart); break;
    }
}

// d_meat: when hc_mode != 0 the caller passes a device buffer of n_rows T values in d_meat (reused as
// the per-row weight vector s); the caller then runs the weighted moment kernel on it.
template <typename T>
int launch_pass2(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, int add_bias, bool weighted,
                 const T* d_beta, const T* d_inv, int hc_mode, T* d_pred, T* d_resid, double* d_sums,
                 double* d_s_rows, double* d_ysums) {
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_feat > kMaxFeatSmall) {
        if (hc_mode >= 2 && !d_inv) return fail(PDS_ERR_INVALID, "HC2 / HC3 need the inverse of X'X");
        const int nb = (int)std::min<int64_t>(std::max<int64_t>((n_rows + kP2Threads - 1) / kP2Threads, 1), (int64_t)ctx->num_cus * 8);
        T* s_rows = hc_mode ? reinterpret_cast<T*>(d_s_rows) : nullptr;
        KernelTimer timer(ctx, kKindPass2);
        if (weighted)
            hipLaunchKernelGGL((pass2_wide_kernel<T, true>), dim3(nb), dim3(kP2Threads), 0, ctx->stream, dc.d_ptrs, n_feat,
                               add_bias ? 1 : 0, n_rows, d_beta, d_pred, d_resid, s_rows, ctx->partials);
        else
            hipLaunchKernelGGL((pass2_wide_kernel<T, false>), dim3(nb), dim3(kP2Threads), 0, ctx->stream, dc.d_ptrs, n_feat,
                               add_bias ? 1 : 0, n_rows, d_beta, d_pred, d_resid, s_rows, ctx->partials);
        hipLaunchKernelGGL(pass2_finalize_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->partials, nb, d_sums);
        bool lev_done = false;
        if constexpr (std::is_same<T, double>::value) {
            // 17 .. 64 f64 features: the leverages as an n x p' x p' product on the matrix cores (leverage_mid.hip);
            // PDS_LEVERAGE_VALU=1 keeps the per-row vector-ALU form below (A/B)
            const char* e = dev_env("PDS_LEVERAGE_VALU");
            if (hc_mode >= 2 && n_feat <= 64 && !(e && e[0] == '1')) {
                PDS_HIP_CHECK(hipGetLastError());
                const int rc = launch_leverage_mid(ctx, dc, n_feat, add_bias ? 1 : 0, n_rows, d_inv, hc_mode, s_rows);
                if (rc == PDS_OK) lev_done = true;
                else if (rc != PDS_ERR_UNSUPPORTED) return rc;
            }
        }
        if (hc_mode >= 2 && !lev_done) {
            const int pp = n_feat + (add_bias ? 1 : 0);
            const size_t lds = (size_t)pp * ((pp + 15) & ~15) * sizeof(double);
            auto regs_form = [&](auto nc_c) {
                constexpr int NC = decltype(nc_c)::value;
                const size_t l2 = (size_t)pp * NC * 16 * sizeof(double);
                if (l2 > 64 * 1024)
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&leverage_scale_wide_regs_kernel<T, NC>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
                hipLaunchKernelGGL((leverage_scale_wide_regs_kernel<T, NC>), dim3(nb), dim3(kP2Threads), l2, ctx->stream, dc.d_ptrs, n_feat,
                                   add_bias ? 1 : 0, n_rows, d_inv, hc_mode, s_rows);
            };
            if (pp <= 32) regs_form(std::integral_constant<int, 2>{});
            else if (pp <= 48) regs_form(std::integral_constant<int, 3>{});
            else if (pp <= 64) regs_form(std::integral_constant<int, 4>{});
            else if (pp <= 80) regs_form(std::integral_constant<int, 5>{});
            else if (lds <= 150 * 1024) {
                if (lds > 64 * 1024)
                    PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&leverage_scale_wide_kernel<T, true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((leverage_scale_wide_kernel<T, true>), dim3(nb), dim3(kP2Threads), lds, ctx->stream, dc.d_ptrs, n_feat,
                                   add_bias ? 1 : 0, n_rows, d_inv, hc_mode, s_rows);
            } else {
                hipLaunchKernelGGL((leverage_scale_wide_kernel<T, false>), dim3(nb), dim3(kP2Threads), 0, ctx->stream, dc.d_ptrs, n_feat,
                                   add_bias ? 1 : 0, n_rows, d_inv, hc_mode, s_rows);
            }
        }
        PDS_HIP_CHECK(hipGetLastError());
        return PDS_OK;
    }
    constexpr int RPL = V16<T>::RPL;
    const int64_t nvec = (n_rows + RPL - 1) / RPL;
    int64_t want = (nvec + kP2Threads - 1) / kP2Threads;
    // two resident blocks per CU (8 waves, like the Gram kernel): every wave streams one contiguous range, all of them at once
    const int nblocks = (int)std::min<int64_t>(std::max<int64_t>(want, 1), (int64_t)ctx->num_cus * PDS_P2_BLOCKS);
    double* partials = ctx->partials;
    double* ypart = d_ysums ? ctx->partials + 4096 : nullptr;  // (2 nblocks <= 1024 doubles each; the context's block is num_cus x 8 records)
    T* s_rows = reinterpret_cast<T*>(d_s_rows);
    KernelTimer timer(ctx, kKindPass2);
    if (weighted)
        launch_p2<T, true>(hc_mode, dim3(nblocks), ctx->stream, dc.d_ptrs, n_feat, add_bias ? 1 : 0, n_rows, d_beta,
                           d_inv, d_pred, d_resid, s_rows, partials, ypart);
    else
        launch_p2<T, false>(hc_mode, dim3(nblocks), ctx->stream, dc.d_ptrs, n_feat, add_bias ? 1 : 0, n_rows, d_beta,
                            d_inv, d_pred, d_resid, s_rows, partials, ypart);
    hipLaunchKernelGGL(pass2_finalize_kernel, dim3(1), dim3(64), 0, ctx->stream, partials, nblocks, d_sums);
    if (ypart) hipLaunchKernelGGL(pass2_finalize_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double*)ypart, nblocks, d_ysums);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

// ---- sum y, sum y^2 in f64 (the f32 report's own var(y): capi_report.hpp) -- per-block partials, fixed-order finish
template <typename T>
__global__ __launch_bounds__(256) void y_sums_kernel(const T* __restrict__ y, int64_t n, double* __restrict__ part, int shifted) {
    __shared__ double sh[2][4];
    double s = 0.0, ss = 0.0;
    // shifted: sums of (y - y[0]) -- the variance from them is the textbook shifted-data form: with y[0] a sample of the data the
    // subtraction ss - s^2 / n no longer cancels when |mean| >> std (timestamps, prices with an offset)
    const double c = (shifted && n > 0) ? (double)y[0] : 0.0;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t a = (int64_t)blockIdx.x * per, b = a + per < n ? a + per : n;
    for (int64_t i = a + threadIdx.x; i < b; i += 256) {
        const double v = (double)y[i] - c;
        s += v;
        ss = fma(v, v, ss);
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o);
        ss += __shfl_down(ss, o);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = s;
        sh[1][threadIdx.x >> 6] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        part[2 * blockIdx.x + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}
__global__ void y_sums_finish_kernel(const double* __restrict__ part, int nblocks, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0, ss = 0.0;
    for (int b = 0; b < nblocks; ++b) {
        s += part[2 * b];
        ss += part[2 * b + 1];
    }
    out[0] = s;
    out[1] = ss;
}
template <typename T>
int launch_y_sums(pds_ctx* ctx, const T* d_y, int64_t n_rows, double* d_out, bool shifted) {
    const int nb = (int)std::min<int64_t>(std::max<int64_t>((n_rows + 4095) / 4096, 1), (int64_t)ctx->num_cus * 4);
    double* part = reinterpret_cast<double*>(ws_take(ctx, sizeof(double) * 2 * (size_t)nb));
    hipLaunchKernelGGL((y_sums_kernel<T>), dim3(nb), dim3(256), 0, ctx->stream, d_y, n_rows, part, shifted ? 1 : 0);
    hipLaunchKernelGGL(y_sums_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double*)part, nb, d_out);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
template int launch_y_sums<double>(pds_ctx*, const double*, int64_t, double*, bool);
template int launch_y_sums<float>(pds_ctx*, const float*, int64_t, double*, bool);

template int launch_pass2<double>(pds_ctx*, const DeviceCols<double>&, int, int64_t, int, bool, const double*,
                                  const double*, int, double*, double*, double*, double*, double*);
template int launch_pass2<float>(pds_ctx*, const DeviceCols<float>&, int, int64_t, int, bool, const float*,
                                 const float*, int, float*, float*, double*, double*, double*);

}  // namespace pds
