// solve_row16_dev.hpp -- normal equations of 17 .. 32 features: FOUR SYSTEMS PER WAVE, one per 16-lane DPP row, two columns per lane.
//
// The one-wave-per-system solver (solve_wave_dev.hpp) spends a whole wave's instruction stream on one 17 .. 32-column system: ~14 000 /
// 20 000 clk each inside the fused grouped kernel of 17 .. 32 features (profiles/r04_grouped_mid_fused.txt) -- more than the stream
// that feeds it.  Here lane t of a DPP row holds columns t and 16 + t of the CENTRED system (all rows: the update keeps both
// triangles, so a lane's own row-K entry IS its multiplier and no lane ever needs another lane's register INDEX), step K broadcasts
// lane (K mod 16)'s entries inside every row with ONE v_mov_b64_dpp each (row_newbcast) and four systems advance per instruction:
// ~800 / 1 450 clk per system at 17 / 32 features.  Same arithmetic as solve_wave_dev.hpp / solve_reg_dev.hpp's chol_step: square-root-free
// L D L', pivot-ratio product as the rank gate (faer_solve_lr_gated's `choleskey` rule, lr_solvers.rs:369-380), suspects marked.
#pragma once
#include "common.hpp"
#include "solve_reg_dev.hpp"

namespace pds {

// a0 / a1: column t / 16 + t of G - s s' / n (+ lambda on the diagonal), rows 0 .. PPC - 1, the centred rhs in [PPC]; zeros beyond p
template <int PPC, int K>
__device__ __forceinline__ void row16_step(double (&a0)[PPC + 1], double (&a1)[PPC + 1], int t, double& invd0, double& invd1, bool& ok) {
    constexpr int S = K / 16, KL = K % 16;  // the pivot column lives in lane KL of every row, column slot S
    const double d = Grp<16>::template bcast<KL>(S == 0 ? a0[K] : a1[K]);
    ok = ok && (d > 0.0);
    double x = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int it = 0; it < PDS_RCP_NEWTON; ++it) x = fma(fma(-d, x, 1.0), x, x);
    // finished columns keep their entries through a zero multiplier (DPP reads from EXEC-disabled lanes are invalid: every lane takes part)
    const double nt0 = (S == 0 && t > KL) ? -(a0[K] * x) : 0.0;
    const double nt1 = (S == 0 || t > KL) ? -(a1[K] * x) : 0.0;
    // (eight rows at a time: left alone the scheduler lifts all of a step's broadcasts to the front -- 64 temporaries beside the 132
    // registers of the system, which does not fit the 256 of a wave that shares its SIMD)
#pragma unroll
    for (int i = K + 1; i <= PPC; ++i) {
        const double b = Grp<16>::template bcast<KL>(S == 0 ? a0[i] : a1[i]);
        if (S == 0) a0[i] = fma(b, nt0, a0[i]);
        a1[i] = fma(b, nt1, a1[i]);
        if ((i - K) % 8 == 0) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (t == KL) {
        if (S == 0) invd0 = x;
        else invd1 = x;
    }
}
// ALL: every one of the PPC steps runs -- columns beyond p are unit columns (diagonal 1, zeros elsewhere: the caller's part), so their
// steps change nothing.  The `K < p` branches of the other form cost a solving wave that shares its SIMD 840 spilled registers at
// PPC = 32 (the values merged behind each branch are copies); without them 86, all of them outside the steps.
template <int PPC, int K, bool ALL = false>
struct Row16Steps {
    static __device__ __forceinline__ void run(double (&a0)[PPC + 1], double (&a1)[PPC + 1], int t, int p, double& i0, double& i1, bool& ok) {
        Row16Steps<PPC, K - 1, ALL>::run(a0, a1, t, p, i0, i1, ok);
        if (ALL || K < p) row16_step<PPC, K>(a0, a1, t, i0, i1, ok);
    }
};
template <int PPC, bool ALL>
struct Row16Steps<PPC, -1, ALL> {
    static __device__ __forceinline__ void run(double (&)[PPC + 1], double (&)[PPC + 1], int, int, double&, double&, bool&) {}
};
// back substitution: w <- w + bcast_M(w) (-a[M] / d_j), a column final after step (its index) + 1
template <int PPC, int M, bool ALL = false>
struct Row16Back {
    static __device__ __forceinline__ void run(const double (&a0)[PPC + 1], const double (&a1)[PPC + 1], int t, int p, double i0, double i1,
                                               double& w0, double& w1) {
        if (ALL || M < p) {
            constexpr int S = M / 16, ML = M % 16;
            const double bm = Grp<16>::template bcast<ML>(S == 0 ? w0 : w1);
            const double c0 = (S == 1 || t < ML) ? -(a0[M] * i0) : 0.0;   // column t < M
            const double c1 = (S == 1 && t < ML) ? -(a1[M] * i1) : 0.0;   // column 16 + t < M
            w0 = fma(bm, c0, w0);
            w1 = fma(bm, c1, w1);
        }
        Row16Back<PPC, M - 1, ALL>::run(a0, a1, t, p, i0, i1, w0, w1);
    }
};
template <int PPC, bool ALL>
struct Row16Back<PPC, 0, ALL> {
    static __device__ __forceinline__ void run(const double (&)[PPC + 1], const double (&)[PPC + 1], int, int, double, double, double&, double&) {}
};

// centring: a[i] -= s_i (s_j / n) for the lane's two columns; s_i comes from lane (i mod 16)'s column sums (slot i / 16)
template <int PPC, int I>
struct Row16Centre {
    static __device__ __forceinline__ void run(double (&a0)[PPC + 1], double (&a1)[PPC + 1], double sj0, double sj1, double m0, double m1) {
        const double si = Grp<16>::template bcast<(I & 15)>(I < 16 ? sj0 : sj1);
        a0[I] = fma(-si, m0, a0[I]);
        a1[I] = fma(-si, m1, a1[I]);
        Row16Centre<PPC, I - 1>::run(a0, a1, sj0, sj1, m0, m1);
    }
};
template <int PPC>
struct Row16Centre<PPC, -1> {
    static __device__ __forceinline__ void run(double (&)[PPC + 1], double (&)[PPC + 1], double, double, double, double) {}
};

// in : a0 / a1 as above, dj0 / dj1 = uncentred diagonal (+ lambda) of the lane's columns (1 where there is none), few = "#Data < #features"
// out: w0 / w1 = the coefficients of columns t / 16 + t; is_null, suspect: uniform inside a 16-lane row (one system)
template <int PPC, bool ALL = false>
__device__ __forceinline__ void row16_ldl_solve(double (&a0)[PPC + 1], double (&a1)[PPC + 1], double dj0, double dj1, int t, int p, bool few,
                                                const SolveRegDev& sp, double& w0, double& w1, bool& is_null, bool& suspect) {
    const bool c0v = t < p, c1v = 16 + t < p;
    is_null = few;
    // a non-positive diagonal entry gates (lr_solvers.rs:341-347)
    if (Grp<16>::sum(((c0v && dj0 <= 0.0) || (c1v && dj1 <= 0.0)) ? 1.0 : 0.0) > 0.0) is_null = true;
    double i0 = 1.0, i1 = 1.0;
    bool ok = true;
    Row16Steps<PPC, PPC - 1, ALL>::run(a0, a1, t, p, i0, i1, ok);
    if (!ok) is_null = true;  // "Not positive-definite -> rank-deficient" (lr_solvers.rs:370-371)
    const double r0 = c0v ? dj0 * i0 : 1.0, r1 = c1v ? dj1 * i1 : 1.0;
    const double grow = grp_prod<16>(r0 * r1);  // prod G_kk / d_k
    if (grow >= sp.inv_tol) is_null = true;
    suspect = solve_suspect(sp, ok, grow, sp.sus_ratio > 0.0 ? grp_max16(fmax(r0, r1)) : 0.0) && !few;
    is_null = is_null || suspect;
    w0 = a0[PPC] * i0;
    w1 = a1[PPC] * i1;
    Row16Back<PPC, PPC - 1, ALL>::run(a0, a1, t, p, i0, i1, w0, w1);
}

}  // namespace pds
