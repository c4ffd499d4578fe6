// solve_wave_dev.hpp -- ONE WAVE PER SYSTEM, the matrix in registers: the device-side core of solve_wave.hip, shared with the
// fused grouped kernel of 17 .. 32 features (grouped_mid.hip, which solves a finished group where its accumulators are).
//
// lane j holds column j of the CENTRED p x p normal equations (a[i] = G_ij - s_i s_j / n, lambda already on the diagonal) and the
// centred right-hand side in a[PPC]; step K broadcasts lane K's entries with v_readlane (compile-time lane: no LDS, no DPP row
// limits) and every lane j > K updates its rows i > K with one FMA each.  Square-root-free L D L' as in solve_reg_dev.hpp's
// chol_step; the rank gate is the product of pivot ratios G_kk / d_k against 1 / tol (faer_solve_lr_gated's `choleskey` rule,
// lr_solvers.rs:369-380, on the uncentred diagonal -- DESIGN.md 4.2); a system next to the gate, gated or broken down is SUSPECT:
// the caller sends it through the reference's default factorisation (the pivoted QR with the log-det gate of solve.hip).
#pragma once
#include "common.hpp"
#include "solve_reg_dev.hpp"

namespace pds {

__device__ __forceinline__ double wave_lane_bcast(double v, int k) {  // k wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_prod64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v *= __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// PPC: compile-time bound of the feature count (a multiple of 8); p <= PPC features, rows / columns beyond p are exact zeros.
// in : a (centred column j, rhs in a[PPC]), dj = uncentred diagonal (+ lambda) of column j (1 for lanes without a column),
//      few = "#Data < #features" (null, never suspect)
// out: w = coefficient j (lanes j < p), is_null, suspect (wave-uniform)
template <int PPC>
__device__ __forceinline__ void wave_ldl_solve(double (&a)[PPC + 1], double dj, int j, int p, bool few, const SolveRegDev& sp, double& w,
                                               bool& is_null, bool& suspect) {
    const bool colv = j < p;
    is_null = few;
    if (__any(colv && dj <= 0.0)) is_null = true;  // a non-positive diagonal entry gates (lr_solvers.rs:341-347)
    double invd = 1.0;
    bool ok = true;
#pragma unroll
    for (int K = 0; K < PPC; ++K) {
        if (K < p) {
            const double d = wave_lane_bcast(a[K], K);
            ok = ok && (d > 0.0);
            double x = __builtin_amdgcn_rcp(d);
#pragma unroll
            for (int it = 0; it < PDS_RCP_NEWTON; ++it) x = fma(fma(-d, x, 1.0), x, x);
            const double nt = (j > K) ? -(a[K] * x) : 0.0;
#pragma unroll
            for (int i = K + 1; i <= PPC; ++i) {
                a[i] = fma(wave_lane_bcast(a[i], K), nt, a[i]);
                // (keep the broadcasts next to their FMAs: hoisted in bulk they needed ~800 scalar registers and spilled)
                if (((i - K) & 7) == 0) __builtin_amdgcn_sched_barrier(0);
            }
            if (j == K) invd = x;
        }
    }
    if (!ok) is_null = true;  // "Not positive-definite -> rank-deficient" (lr_solvers.rs:370-371)
    const double ratio = colv ? dj * invd : 1.0;
    const double grow = wave_prod64(ratio);  // prod G_kk / L_kk^2
    if (grow >= sp.inv_tol) is_null = true;
    suspect = solve_suspect(sp, ok, grow, sp.sus_ratio > 0.0 ? wave_max64(ratio) : 0.0) && !few;
    is_null = is_null || suspect;
    // ---- back substitution: w <- w + bcast_M(w) (-a[M] / d_j), lane j final after step j + 1
    w = a[PPC] * invd;
#pragma unroll
    for (int Mi = PPC - 1; Mi >= 1; --Mi) {
        if (Mi < p) {
            const double c = (j < Mi) ? -(a[Mi] * invd) : 0.0;
            w = fma(wave_lane_bcast(w, Mi), c, w);
        }
    }
}

}  // namespace pds
