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
#include "common.hpp"

namespace pds {

struct SolveRegDev {
    int p, pp, bias, lambda_on_bias, gate_on;
    double lambda, ln_tol;
};

template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ double dpp_mov(double old, double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int olo = __double2loint(old), ohi = __double2hiint(old);
    lo = __builtin_amdgcn_update_dpp(olo, lo, CTRL, 0xf, BANK, false);
    hi = __builtin_amdgcn_update_dpp(ohi, hi, CTRL, 0xf, BANK, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

constexpr int kRor8 = 0x128, kRor4 = 0x124, kRor2 = 0x122, kRor1 = 0x121;
constexpr int kHalfMirror = 0x141, kXor1 = 0xB1 /*quad_perm[1,0,3,2]*/, kXor2 = 0x4E /*quad_perm[2,3,0,1]*/;
constexpr int kNewBcast = 0x150;

template <int LPS>
struct Grp;

template <>
struct Grp<16> {
    static __device__ __forceinline__ double sum(double v) {
        v += dpp_mov<kRor8>(0.0, v);
        v += dpp_mov<kRor4>(0.0, v);
        v += dpp_mov<kRor2>(0.0, v);
        v += dpp_mov<kRor1>(0.0, v);
        return v;
    }
    template <int C>
    static __device__ __forceinline__ void amax_step(double& v, int& idx) {
        const double ov = dpp_mov<C>(0.0, v);
        const int oi = dpp_mov_i<C>(idx);
        if (ov > v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
    static __device__ __forceinline__ void argmax(double& v, int& idx) {
        amax_step<kRor8>(v, idx);
        amax_step<kRor4>(v, idx);
        amax_step<kRor2>(v, idx);
        amax_step<kRor1>(v, idx);
    }
    template <int K>
    static __device__ __forceinline__ double bcast(double v) {
        return dpp_mov<kNewBcast + K>(0.0, v);
    }
};

template <>
struct Grp<8> {
    static __device__ __forceinline__ double sum(double v) {
        v += dpp_mov<kHalfMirror>(0.0, v);
        v += dpp_mov<kXor1>(0.0, v);
        v += dpp_mov<kXor2>(0.0, v);
        return v;
    }
    template <int C>
    static __device__ __forceinline__ void amax_step(double& v, int& idx) {
        const double ov = dpp_mov<C>(0.0, v);
        const int oi = dpp_mov_i<C>(idx);
        if (ov > v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
    static __device__ __forceinline__ void argmax(double& v, int& idx) {
        amax_step<kHalfMirror>(v, idx);
        amax_step<kXor1>(v, idx);
        amax_step<kXor2>(v, idx);
    }
    template <int K>
    static __device__ __forceinline__ double bcast(double v) {
        // two half-rows: banks 0-1 take lane K, banks 2-3 take lane 8+K of the 16-lane DPP row
        double r = dpp_mov<kNewBcast + K, 0x3>(0.0, v);
        return dpp_mov<kNewBcast + 8 + K, 0xC>(r, v);
    }
};

template <>
struct Grp<4> {
    static __device__ __forceinline__ double sum(double v) {
        v += dpp_mov<kXor1>(0.0, v);
        v += dpp_mov<kXor2>(0.0, v);
        return v;
    }
    template <int C>
    static __device__ __forceinline__ void amax_step(double& v, int& idx) {
        const double ov = dpp_mov<C>(0.0, v);
        const int oi = dpp_mov_i<C>(idx);
        if (ov > v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
    static __device__ __forceinline__ void argmax(double& v, int& idx) {
        amax_step<kXor1>(v, idx);
        amax_step<kXor2>(v, idx);
    }
    template <int K>
    static __device__ __forceinline__ double bcast(double v) {
        // quad_perm [K,K,K,K]
        return dpp_mov<(K) | (K << 2) | (K << 4) | (K << 6)>(0.0, v);
    }
};

__device__ __forceinline__ double bperm_f64(int src_lane, double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, lo);
    hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, hi);
    return __hiloint2double(hi, lo);
}

// one Householder step, K static.  a[]: this lane's column, b[]: rhs replica, rd: captured R_jj.
template <int LPS, int K>
__device__ __forceinline__ void qr_step(double (&a)[LPS], double (&b)[LPS], int j, int lane, int pp, int& pj,
                                        double& rd) {
    // ---- pivot: largest remaining column norm (squared), lowest index on ties
    double nrm = 0.0;
#pragma unroll
    for (int i = K; i < LPS; ++i) nrm = fma(a[i], a[i], nrm);
    double best = (j >= K && j < pp) ? nrm : -1.0;
    if (!(best == best)) best = -1.0;
    int pv = j;
    Grp<LPS>::argmax(best, pv);
    if (best < 0.0) pv = K;
    if (__any(pv != K)) {  // exchange columns K <-> pv inside each sub-group
        const int src = (j == K) ? pv : ((j == pv) ? K : j);
        const int src_lane = (lane - j) + src;
#pragma unroll
        for (int i = 0; i < LPS; ++i) a[i] = bperm_f64(src_lane, a[i]);
        pj = __builtin_amdgcn_ds_bpermute(src_lane << 2, pj);
    }
    // ---- reflector from column K (now in lane K): v = x - beta e_K
    double v[LPS];
#pragma unroll
    for (int i = K; i < LPS; ++i) v[i] = Grp<LPS>::template bcast<K>(a[i]);
    const double alpha = v[K];
    double xn2 = 0.0;
#pragma unroll
    for (int i = K + 1; i < LPS; ++i) xn2 = fma(v[i], v[i], xn2);
    double rkk = alpha;
    if (xn2 != 0.0) {
        double beta = sqrt(fma(alpha, alpha, xn2));
        if (alpha > 0.0) beta = -beta;
        rkk = beta;
        v[K] = alpha - beta;
        const double hinv = 1.0 / (beta * (beta - alpha));  // 2 / (v'v)
        // own column (only columns to the right of K change; lane K itself becomes [.., beta, 0..])
        double w = 0.0;
#pragma unroll
        for (int i = K; i < LPS; ++i) w = fma(v[i], a[i], w);
        w *= hinv;
        if (j > K) {
#pragma unroll
            for (int i = K; i < LPS; ++i) a[i] = fma(-w, v[i], a[i]);
        }
        double wb = 0.0;
#pragma unroll
        for (int i = K; i < LPS; ++i) wb = fma(v[i], b[i], wb);
        wb *= hinv;
#pragma unroll
        for (int i = K; i < LPS; ++i) b[i] = fma(-wb, v[i], b[i]);
    }
    if (j == K) {
        a[K] = rkk;
        rd = rkk;
    }
}

template <int LPS, int K>
struct QrSteps {
    static __device__ __forceinline__ void run(double (&a)[LPS], double (&b)[LPS], int j, int lane, int pp, int& pj,
                                               double& rd) {
        QrSteps<LPS, K - 1>::run(a, b, j, lane, pp, pj, rd);
        if (K < pp) qr_step<LPS, K>(a, b, j, lane, pp, pj, rd);
    }
};
template <int LPS>
struct QrSteps<LPS, -1> {
    static __device__ __forceinline__ void run(double (&)[LPS], double (&)[LPS], int, int, int, int&, double&) {}
};

// row-parallel back substitution: lane j holds R[i][j] = a[i] (i <= j) and ends with z_j
template <int LPS, int I>
struct BackSub {
    static __device__ __forceinline__ void run(const double (&a)[LPS], const double (&b)[LPS], int j, int pp, double rinv,
                                               double& zj) {
        if (I < pp) {
            const double t = (j > I && j < pp) ? a[I] * zj : 0.0;
            const double s = Grp<LPS>::sum(t);
            const double ri = Grp<LPS>::template bcast<I>(rinv);
            const double zi = (b[I] - s) * ri;
            if (j == I) zj = zi;
        }
        BackSub<LPS, I - 1>::run(a, b, j, pp, rinv, zj);
    }
};
template <int LPS>
struct BackSub<LPS, -1> {
    static __device__ __forceinline__ void run(const double (&)[LPS], const double (&)[LPS], int, int, double, double&) {}
};

template <typename T, int LPS>
__global__ __launch_bounds__(256) void solve_reg_kernel(const T* __restrict__ moments, int64_t n_sys, SolveRegDev sp,
                                                        T* __restrict__ coeffs, uint8_t* __restrict__ flags,
                                                        const int64_t* __restrict__ rows_per_sys) {
    constexpr int SPW = 64 / LPS;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPS, j = lane % LPS;
    const int p = sp.p, pp = sp.pp, q = p + 2;
    const int64_t waves_total = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (int64_t base = wave_id * SPW; base < n_sys; base += waves_total * SPW) {
        const int64_t sys = base + sub;
        const bool live = sys < n_sys;
        const T* M = moments + (live ? sys : 0) * (int64_t)(q * q);
        double a[LPS], b[LPS];
        const bool colv = j < pp;
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            a[i] = (colv && i < pp) ? (double)M[i + j * q] : 0.0;
            b[i] = (i < pp) ? (double)M[i + (p + 1) * q] : 0.0;
        }
        double dj = colv ? (double)M[j + j * q] : 1.0;
        if (sp.lambda > 0.0 && colv && (j < p || sp.lambda_on_bias)) {
            dj += sp.lambda;
#pragma unroll
            for (int i = 0; i < LPS; ++i)
                if (i == j) a[i] += sp.lambda;
        }
        bool is_null = false;
        if (rows_per_sys && live && rows_per_sys[sys + 1] - rows_per_sys[sys] < pp) is_null = true;
        double ln_den = 0.0;
        if (sp.gate_on) {
            const double bad = Grp<LPS>::sum((colv && !(dj > 0.0)) ? 1.0 : 0.0);  // `d <= 0` (NaN passes, as in Rust)
            const double badn = Grp<LPS>::sum((colv && dj != dj) ? 1.0 : 0.0);
            if (bad - badn > 0.0) is_null = true;
            ln_den = Grp<LPS>::sum(colv ? log(dj) : 0.0);
        }
        int pj = j;
        double rd = 1.0;
        QrSteps<LPS, LPS - 1>::run(a, b, j, lane, pp, pj, rd);
        const double ln_det = Grp<LPS>::sum(colv ? log(fabs(rd)) : 0.0);
        if (sp.gate_on && !is_null && (ln_det - ln_den <= sp.ln_tol)) is_null = true;
        const double rinv = 1.0 / rd;
        double zj = 0.0;
        BackSub<LPS, LPS - 1>::run(a, b, j, pp, rinv, zj);
        if (live && colv) coeffs[sys * (int64_t)pp + pj] = is_null ? (T)__builtin_nan("") : (T)zj;
        if (live && j == 0 && flags) flags[sys] = is_null ? 1 : 0;
    }
}

template <typename T, int LPS>
static int launch_lps(pds_ctx* ctx, const T* d_moments, int64_t n_sys, const SolveRegDev& sd, T* d_coeffs,
                      uint8_t* d_flags, const int64_t* d_rows) {
    constexpr int SPW = 64 / LPS;
    const int64_t waves = (n_sys + SPW - 1) / SPW;
    int64_t nb = (waves + 3) / 4;
    nb = std::min<int64_t>(std::max<int64_t>(nb, 1), (int64_t)ctx->num_cus * 16);
    KernelTimer timer(ctx, kKindSolve);
    hipLaunchKernelGGL((solve_reg_kernel<T, LPS>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_moments, n_sys, sd,
                       d_coeffs, d_flags, d_rows);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

// returns PDS_ERR_UNSUPPORTED (without setting an error) when the caller should use the LDS kernel
template <typename T>
int launch_solve_reg(pds_ctx* ctx, const T* d_moments, int64_t n_sys, const SolveParams& sp, T* d_coeffs,
                     uint8_t* d_flags, const int64_t* d_rows_per_sys) {
    SolveRegDev sd;
    sd.p = sp.p;
    sd.bias = sp.add_bias ? 1 : 0;
    sd.pp = sp.p + sd.bias;
    sd.lambda_on_bias = sp.lambda_on_bias;
    sd.lambda = sp.lambda;
    sd.gate_on = sp.gate_tol > 0.0 ? 1 : 0;
    sd.ln_tol = sd.gate_on ? std::log(sp.gate_tol) : 0.0;
    if (sd.pp <= 4) return launch_lps<T, 4>(ctx, d_moments, n_sys, sd, d_coeffs, d_flags, d_rows_per_sys);
    if (sd.pp <= 8) return launch_lps<T, 8>(ctx, d_moments, n_sys, sd, d_coeffs, d_flags, d_rows_per_sys);
    return launch_lps<T, 16>(ctx, d_moments, n_sys, sd, d_coeffs, d_flags, d_rows_per_sys);
}

template int launch_solve_reg<double>(pds_ctx*, const double*, int64_t, const SolveParams&, double*, uint8_t*,
                                      const int64_t*);
template int launch_solve_reg<float>(pds_ctx*, const float*, int64_t, const SolveParams&, float*, uint8_t*,
                                     const int64_t*);

}  // namespace pds
