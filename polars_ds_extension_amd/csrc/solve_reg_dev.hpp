// solve_reg_dev.hpp -- device-side building blocks of the register-resident pivoted-QR solve (DPP
// sub-wave groups), shared by solve_reg.hip and grouped_fused.hip.  See solve_reg.hip for the design notes.
#pragma once
#include <cstdlib>
#include "common.hpp"

namespace pds {

struct SolveRegDev {
    int p, pp, bias, lambda_on_bias, gate_on;
    double lambda, ln_tol;
    double inv_tol;  // 1 / gate_tol (the Cholesky gate compares a product of pivot ratios)
    // Fused grouped kernel with solver = "qr" (the reference's default col_piv_qr): a system whose pivot-ratio product
    // reaches sus_tol = sqrt(1 / gate_tol) -- rel. determinant within [tol, sqrt(tol)], every gated one, every breakdown --
    // is not answered by the Cholesky at all but marked for the pivoted-QR kernel, so that the null decision next to the
    // threshold and the coefficients of marginal systems come from the reference's own factorisation.  0 = off.
    double sus_tol = 0.0;
    // The wave / row16 solvers (17 .. 64 features) mark less (round 4).  The reference's gate statistic, sum ln |R_ii| - sum ln G_ii of the
    // pivoted QR, is ln det G - sum ln G_ii whatever the pivoting: MINUS the log of the pivot-ratio product these solvers form, so the
    // null decision only needs the reference's own arithmetic where the two roundings could disagree -- a product within sus_band
    // (relative) of 1 / tol -- and the coefficients only where a single pivot has lost sus_ratio of its diagonal (conditioning; a
    // product of many moderate ratios, e.g. 64 random columns over 100 rows: 6.6e11 from ratios below 3, is no loss of accuracy).
    // sus_ratio = 0: the product rule above.
    double sus_ratio = 0.0, sus_band = 0.0;
};
// (host) the single-ratio bound: 1e5 unless PDS_SUSPECT_RATIO says otherwise, read per call (0: the product rule, for A/B)
inline double solve_suspect_ratio() {
    const char* e = dev_env("PDS_SUSPECT_RATIO");
    return e ? std::atof(e) : 1e5;
}
// suspect rule of the wave / row16 solvers (grow: product of the pivot ratios, rmax: the largest of them)
__device__ __forceinline__ bool solve_suspect(const SolveRegDev& sp, bool ok, double grow, double rmax) {
    if (!(sp.sus_tol > 0.0)) return false;
    if (!(sp.sus_ratio > 0.0)) return !ok || !(grow < sp.sus_tol);
    return !ok || !(rmax < sp.sus_ratio) || (grow >= sp.inv_tol * (1.0 - sp.sus_band) && grow <= sp.inv_tol * (1.0 + sp.sus_band));
}

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
// Cholesky (llt(Side::Lower), lr_solvers.rs:288,369) in the same lane-per-column register layout, carried out as
// the square-root-free L D L' it is equivalent to.  The full symmetric matrix is kept (lane j = column j,
// a[i] = G_ij), so the multiplier of column j in step K is its OWN row-K entry over the pivot -- no transpose
// traffic -- and the rhs rides along as one extra ROW (a[LPS] = c_j).  After the elimination lane j holds
// d_j = L_jj^2 (captured as 1 / d_j), the unscaled column L_ij L_jj in a[i] (i > j) and L_jj y_j in a[LPS], so
//     beta_j = a[LPS] / d_j - sum_{M > j} (a[M] / d_j) beta_M
// needs neither a square root nor 1 / L_jj.  A non-positive pivot means "not positive definite" (:370-371).
// Instruction shape (the solve is a latency chain, every dependent instruction costs ~10 clk of a wave's time):
//   * step K: pivot broadcast, v_rcp_f64 + two Newton steps (4 dependent FMAs; the rsqrt form needed 6 + 2),
//     one multiply, then ONE v_fmac_f64_dpp per remaining row;
//   * back substitution: w <- w + bcast_M(w) * (-a[M] / d_j) for M = p'-1 .. 1 is the same single instruction per
//     step on a running vector w whose lane j is final (= beta_j) once step j + 1 has passed -- no selects;
//   * the rank gate sum ln L_kk^2 - sum ln G_kk <= ln tol (:372-380) is taken as prod (G_kk / d_k) >= 1 / tol: the
//     factors are >= 1, an overflow lands on the right side, and no logarithm sits on the chain.
// ---------------------------------------------------------------------------------------------
// a + bcast_K(a) * m over the sub-group's lanes
template <int LPS, int K>
__device__ __forceinline__ void bcast_fmac(double& a, double m) {
    if constexpr (LPS == 16) {
        // ONE instruction: the DP ALU takes row_newbcast on its first source (the compiler only ever emits
        // v_mov_b64_dpp + v_fma_f64 for this).  A DPP read needs two wait states after a VALU write of the same
        // register and the compiler cannot see into the asm -- it does place register copies directly in front of it
        // (observed) -- so the wait states travel with the instruction.
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(m), "n"(K));
    } else {
        a = fma(Grp<LPS>::template bcast<K>(a), m, a);
    }
}

template <int LPS, int K>
__device__ __forceinline__ void chol_step(double (&a)[LPS + 1], int j, double& invd, bool& ok) {
    const double d = Grp<LPS>::template bcast<K>(a[K]);
    ok = ok && (d > 0.0);
    double x = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int it = 0; it < PDS_RCP_NEWTON; ++it) x = fma(fma(-d, x, 1.0), x, x);
    // lanes <= K keep their finished columns: a zero multiplier instead of a select per element (DPP reads from
    // EXEC-disabled lanes are invalid, so every lane takes part)
    const double nt = (j > K) ? -(a[K] * x) : 0.0;
#pragma unroll
    for (int i = K + 1; i <= LPS; ++i) bcast_fmac<LPS, K>(a[i], nt);
    if (j == K) invd = x;
}

template <int LPS, int K>
struct CholSteps {
    static __device__ __forceinline__ void run(double (&a)[LPS + 1], int j, int pp, double& invd, bool& ok) {
        CholSteps<LPS, K - 1>::run(a, j, pp, invd, ok);
        if (K < pp) chol_step<LPS, K>(a, j, invd, ok);
    }
};
template <int LPS>
struct CholSteps<LPS, -1> {
    static __device__ __forceinline__ void run(double (&)[LPS + 1], int, int, double&, bool&) {}
};

// rows / columns beyond p' are exact zeros (see the callers' loads), so their steps add zeros: no p' guard
template <int LPS, int M>
struct CholBack {
    static __device__ __forceinline__ void run(const double (&c)[LPS], double& w) {
        bcast_fmac<LPS, M>(w, c[M]);
        CholBack<LPS, M - 1>::run(c, w);
    }
};
template <int LPS>
struct CholBack<LPS, 0> {
    static __device__ __forceinline__ void run(const double (&)[LPS], double&) {}
};

__device__ __forceinline__ double grp_max16(double v) {
    v = fmax(v, dpp_mov<kRor8>(0.0, v));
    v = fmax(v, dpp_mov<kRor4>(0.0, v));
    v = fmax(v, dpp_mov<kRor2>(0.0, v));
    v = fmax(v, dpp_mov<kRor1>(0.0, v));
    return v;
}
template <int LPS>
__device__ __forceinline__ double grp_prod(double v);
template <>
__device__ __forceinline__ double grp_prod<16>(double v) {
    v *= dpp_mov<kRor8>(0.0, v);
    v *= dpp_mov<kRor4>(0.0, v);
    v *= dpp_mov<kRor2>(0.0, v);
    v *= dpp_mov<kRor1>(0.0, v);
    return v;
}
template <>
__device__ __forceinline__ double grp_prod<8>(double v) {
    v *= dpp_mov<kHalfMirror>(0.0, v);
    v *= dpp_mov<kXor1>(0.0, v);
    v *= dpp_mov<kXor2>(0.0, v);
    return v;
}
template <>
__device__ __forceinline__ double grp_prod<4>(double v) {
    v *= dpp_mov<kXor1>(0.0, v);
    v *= dpp_mov<kXor2>(0.0, v);
    return v;
}

// a[0..LPS) = column j of G (+lambda), a[LPS] = c_j, dj = G_jj.  beta_j is the coefficient of column j.
template <int LPS>
__device__ __forceinline__ void chol_core(double (&a)[LPS + 1], double dj, int j, const SolveRegDev& sp, bool& is_null,
                                          double& beta, bool* suspect = nullptr) {
    const int pp = sp.pp;
    const bool colv = j < pp;
    if (sp.gate_on) {
        // a non-positive diagonal entry gates (NaN passes, as in Rust)   lr_solvers.rs:341-347
        const double bad = Grp<LPS>::sum((colv && dj <= 0.0) ? 1.0 : 0.0);
        if (bad > 0.0) is_null = true;
    }
    double invd = 1.0;
    bool ok = true;
    CholSteps<LPS, LPS - 1>::run(a, j, pp, invd, ok);
    if (!ok) is_null = true;  // "Not positive-definite -> rank-deficient" (lr_solvers.rs:370-371)
    if (sp.gate_on) {
        const double grow = grp_prod<LPS>(colv ? dj * invd : 1.0);  // prod G_kk / L_kk^2
        if (grow >= sp.inv_tol) is_null = true;
        if (suspect) *suspect = sp.sus_tol > 0.0 && (!ok || !(grow < sp.sus_tol));
    }
    double c[LPS];
#pragma unroll
    for (int m = 1; m < LPS; ++m) c[m] = (j < m) ? -(a[m] * invd) : 0.0;
    double w = a[LPS] * invd;
    CholBack<LPS, LPS - 1>::run(c, w);
    beta = w;
}

}  // namespace pds
