// solve_reg_dev.hpp -- device-side building blocks of the register-resident pivoted-QR solve (DPP
// sub-wave groups), shared by solve_reg.hip and grouped_fused.hip.  See solve_reg.hip for the design notes.
#pragma once
#include "common.hpp"

namespace pds {

struct SolveRegDev {
    int p, pp, bias, lambda_on_bias, gate_on;
    double lambda, ln_tol;
};

// Cross-lane moves of a double.  The f64 overload of update_dpp matters: with row_newbcast it is ONE v_mov_b64_dpp
// (the only DPP control the DP ALU takes), where moving the halves as two ints cost 2 x (v_mov_b32 to seed `old` +
// v_mov_b32_dpp).  Unmasked moves use bound_ctrl (lanes without a source read 0), which also spares the seeding
// moves of the two-instruction forms (row_ror, quad_perm, row_half_mirror); `old` is only honoured with a bank mask.
template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ double dpp_mov(double old, double v) {
    if constexpr (BANK == 0xf) {
        (void)old;  // every unmasked caller passes 0
        return __builtin_amdgcn_update_dpp(0.0, v, CTRL, 0xf, 0xf, true);
    } else {
        return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, BANK, false);
    }
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
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


// The whole solve of one system held by a sub-wave group: a[] = column j of G (+lambda applied), b[] = rhs
// replica, dj = G_jj.  Returns z_j (coefficient of original column pj) and the null flag.
template <int LPS>
__device__ __forceinline__ void solve_core(double (&a)[LPS], double (&b)[LPS], double dj, int j, int lane,
                                           const SolveRegDev& sp, bool& is_null, int& pj, double& zj) {
    const int pp = sp.pp;
    const bool colv = j < pp;
    double ln_den = 0.0;
    if (sp.gate_on) {
        const double bad = Grp<LPS>::sum((colv && !(dj > 0.0)) ? 1.0 : 0.0);  // `d <= 0` (NaN passes, as in Rust)
        const double badn = Grp<LPS>::sum((colv && dj != dj) ? 1.0 : 0.0);
        if (bad - badn > 0.0) is_null = true;
        ln_den = Grp<LPS>::sum(colv ? log(dj) : 0.0);
    }
    pj = j;
    double rd = 1.0;
    QrSteps<LPS, LPS - 1>::run(a, b, j, lane, pp, pj, rd);
    const double ln_det = Grp<LPS>::sum(colv ? log(fabs(rd)) : 0.0);
    if (sp.gate_on && !is_null && (ln_det - ln_den <= sp.ln_tol)) is_null = true;
    const double rinv = 1.0 / rd;
    zj = 0.0;
    BackSub<LPS, LPS - 1>::run(a, b, j, pp, rinv, zj);
}

// ---------------------------------------------------------------------------------------------
// Cholesky (llt(Side::Lower), lr_solvers.rs:288,369) in the same lane-per-column register layout.
// The full symmetric matrix is kept (lane j = column j, a[i] = G_ij), so the multiplier l_jK of column j is
// its OWN row-K entry -- no transpose traffic -- and the rhs rides along as one extra ROW (a[LPS] = c_j):
// after the factorisation lane j holds y_j = (L^-1 c)_j for free.  ~4x fewer instructions and a third of
// the registers of the pivoted QR; a non-positive pivot means "not positive definite".
// ---------------------------------------------------------------------------------------------
template <int LPS, int K>
__device__ __forceinline__ void chol_step(double (&a)[LPS + 1], int j, int pp, double& rinv_j, double& dcap,
                                          bool& ok) {
    if constexpr (LPS == 16) asm volatile("s_nop 1" ::: "memory");  // see the inline-asm update below
    const double d = Grp<LPS>::template bcast<K>(a[K]);
    ok = ok && (d > 0.0);
    // 1/sqrt(d): hardware estimate (v_rsq_f64) + two Newton steps -- a third of the instructions of sqrt + divide,
    // accurate to an ulp or two, which is all the elimination multipliers need (and fails the same way for d <= 0)
    double r = __builtin_amdgcn_rsq(d);
    r = r * fma(-0.5 * d * r, r, 1.5);
    r = r * fma(-0.5 * d * r, r, 1.5);
    // lanes <= K keep their finished columns (their a[i], i > K, are the unscaled l_iK): a zero multiplier instead
    // of a select per element.  DPP reads from EXEC-disabled lanes are invalid, so every lane takes part.
    const double t = (j > K) ? a[K] * r * r : 0.0;  // l_jK / l_KK
    if constexpr (LPS == 16) {
        // a[i] += bcast_K(a[i]) * (-t) as ONE instruction per row: the DP ALU takes row_newbcast on its first source
        // (the compiler only ever emits v_mov_b64_dpp + v_fma_f64 for this).  The 2-wait-state rule between a VALU
        // write and a DPP read of the same register is ours to keep inside inline asm: a[i] was last written one
        // whole elimination step ago, and chol_step opens with an s_nop for the shortest steps.
        const double nt = -t;
#pragma unroll
        for (int i = K + 1; i <= LPS; ++i)
            asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(nt), "n"(K));
    } else {
#pragma unroll
        for (int i = K + 1; i <= LPS; ++i) {
            const double v = Grp<LPS>::template bcast<K>(a[i]);
            a[i] = fma(-v, t, a[i]);
        }
    }
    if (j == K) {
        rinv_j = r;
        dcap = d;
    }
}

template <int LPS, int K>
struct CholSteps {
    static __device__ __forceinline__ void run(double (&a)[LPS + 1], int j, int pp, double& rinv_j, double& dcap,
                                               bool& ok) {
        CholSteps<LPS, K - 1>::run(a, j, pp, rinv_j, dcap, ok);
        if (K < pp) chol_step<LPS, K>(a, j, pp, rinv_j, dcap, ok);
    }
};
template <int LPS>
struct CholSteps<LPS, -1> {
    static __device__ __forceinline__ void run(double (&)[LPS + 1], int, int, double&, double&, bool&) {}
};

template <int LPS, int M>
struct CholBack {
    static __device__ __forceinline__ void run(const double (&a)[LPS + 1], int j, int pp, double yj, double rinv_j,
                                               double& acc, double& beta) {
        if (M < pp) {
            const double cand = (yj - acc) * rinv_j;
            const double bm = Grp<LPS>::template bcast<M>(cand);
            if (j == M) beta = bm;
            if (j < M) acc = fma(a[M] * rinv_j, bm, acc);  // l_Mj beta_M, l_Mj = a[M] / l_jj
        }
        CholBack<LPS, M - 1>::run(a, j, pp, yj, rinv_j, acc, beta);
    }
};
template <int LPS>
struct CholBack<LPS, -1> {
    static __device__ __forceinline__ void run(const double (&)[LPS + 1], int, int, double, double, double&, double&) {}
};

// a[0..LPS) = column j of G (+lambda), a[LPS] = c_j, dj = G_jj.  beta_j is the coefficient of column j.
template <int LPS>
__device__ __forceinline__ void chol_core(double (&a)[LPS + 1], double dj, int j, const SolveRegDev& sp, bool& is_null,
                                          double& beta) {
    const int pp = sp.pp;
    const bool colv = j < pp;
    double ln_den = 0.0;
    if (sp.gate_on) {
        const double bad = Grp<LPS>::sum((colv && !(dj > 0.0)) ? 1.0 : 0.0);
        const double badn = Grp<LPS>::sum((colv && dj != dj) ? 1.0 : 0.0);
        if (bad - badn > 0.0) is_null = true;
        ln_den = Grp<LPS>::sum(colv ? log(dj) : 0.0);
    }
    double rinv_j = 1.0, dcap = 1.0;
    bool ok = true;
    CholSteps<LPS, LPS - 1>::run(a, j, pp, rinv_j, dcap, ok);
    if (!ok) is_null = true;  // "Not positive-definite -> rank-deficient" (lr_solvers.rs:370-371)
    if (sp.gate_on && !is_null) {
        const double ln_det = Grp<LPS>::sum(colv ? log(dcap) : 0.0);  // = 2 sum ln L_kk
        if (ln_det - ln_den <= sp.ln_tol) is_null = true;
    }
    const double yj = a[LPS] * rinv_j;
    double acc = 0.0;
    beta = 0.0;
    CholBack<LPS, LPS - 1>::run(a, j, pp, yj, rinv_j, acc, beta);
}

}  // namespace pds
