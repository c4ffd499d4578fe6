// moments.hip -- the Gram build: one streaming pass over the column buffers producing the augmented
// moment matrix A = Z'Z, Z = [x_0 .. x_{p-1} | 1 | y]  (replaces get_xtx_with_lambda + build_xty,
// /root/reference/src/linear/lr/lr_solvers.rs:183-211, 262-278, and the column sums of :483-484).
//
// gfx950 design (DESIGN.md section "Gram kernel"):
//   * the job is HBM-bound (3.8 flop/B at p=16 f64), so every input element is read exactly once;
//   * each wave owns a private LDS tile: it reads 16 B per lane straight down one column (1 KiB
//     fully coalesced per load instruction), parks the column in LDS, and re-reads it in the MFMA
//     operand layout (feature = lane & 15, row slot = lane >> 4).  For a Gram matrix the A and B
//     operands of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 are the SAME register;
//   * y, the bias (column sums) and the weights ride along on the VALU (it idles under the MFMA);
//   * LDS column stride 1040 B makes the ds_read_b64 operand fetch conflict free;
//   * no atomics: per-block partials + a fixed-order finalize kernel => run-to-run bit reproducible.
#include <type_traits>

#include "moments_dev.hpp"

#ifndef PDS_LEV_NB
#define PDS_LEV_NB 2  // (4: same speed, 6 spilled VGPRs + 28 B of scratch at 256 registers; round 6 A/B) 16-row blocks of a tile whose leverage matrix steps are interleaved (moments_small_kernel, LEVM)
#endif

namespace pds {

// link / variance functions of the GLM (link_functions.rs:5-77): 0 identity / gaussian, 1 log / poisson, 2 logit / binomial,
// 3 inverse / gamma -- evaluated in T, like the reference's `T: RealField + Float`
template <typename T>
__device__ __forceinline__ T glm_link(int link, T mu) {
    switch (link) {
        case 1: return (T)log(mu);
        case 2: return (T)log(mu / (T(1) - mu));
        case 3: return T(1) / mu;
        default: return mu;
    }
}
template <typename T>
__device__ __forceinline__ T glm_inv(int link, T eta) {
    switch (link) {
        case 1: return (T)exp(eta);
        case 2: { const T e = (T)exp(eta); return e / (T(1) + e); }
        case 3: return T(1) / eta;
        default: return eta;
    }
}
template <typename T>
__device__ __forceinline__ T glm_deriv(int link, T mu) {
    switch (link) {
        case 1: return T(1) / mu;
        case 2: return T(1) / (mu * (T(1) - mu));
        case 3: { const T r = T(1) / mu; return -(r * r); }
        default: return T(1);
    }
}
template <typename T>
__device__ __forceinline__ T glm_var(int variance, T mu) {
    switch (variance) {
        case 1: return mu;
        case 2: return mu * (T(1) - mu);
        case 3: return mu * mu;
        default: return T(1);
    }
}

// =============================================================================================
// single big system: grid-stride over 128-row (f64) / 256-row (f32) tiles, register prefetch of the
// next tile while the matrix core chews the current one.
// =============================================================================================
// P16: exactly 16 features, known at compile time -- the per-column `c < p` scalar branches of the tile load / store fold away
// P2 (0 = off): p <= P2 <= 8 features, 16 / P2 row slabs per matrix instruction (consume_tile_pack)
// WM = 3: one step of iteratively re-weighted least squares (faer_irls, glm_solvers.rs:293-316) as ONE Gram pass: from the
// row it has just loaded the lane forms eta = x . beta_prev, mu = g^-1(eta), the weight 1 / (g'(mu)^2 V(mu)) and the working
// response z = eta + g'(mu) (y - mu), which REPLACES y in the tile -- the reference's n-long mu / eta / weights / d_mu vectors
// never exist, an IRLS iteration reads X and y once and writes nothing.  ia.init: the first iteration starts from
// mu0 = (y + mean) / 2 (binomial: (y + 0.5) / 2), eta0 = g(mu0) (:263-290).
// WM: 0 = unweighted, 1 = a weight column, 2 = the weight of a row is its squared residual under `beta` (the HC0 / HC1
// "meat" X' diag(e^2) X of pl_lin_reg_report, linear_regression.rs:880-892, in the SAME pass that forms the residuals: the
// lane that loaded a row holds all of its features, so e_i = y_i - x_i . beta costs p FMAs before the tile goes to LDS, the
// per-row weight vector never reaches HBM, and sum(w) of the record is the residual sum of squares)
template <typename T, int WM, bool P16, int P2>
__global__ __launch_bounds__(256, 2) void moments_small_kernel(const T* const* __restrict__ cols, int p_arg,
                                                               int64_t n, double* __restrict__ partials,
                                                               const T* __restrict__ beta, int bias, IrlsArgs ia) {
    static_assert(!(P16 && P2), "one or the other");
    constexpr bool WEIGHTED = WM != 0;   // the LDS tile carries a weight column
    constexpr bool LOADW = WM == 1;      // ... that comes from memory
    const int p = P16 ? 16 : p_arg;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RPL = Tile<T>::RPL;
    constexpr int TR = 64 * RPL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* wl = smem + wave * kWaveLds;

    // unused feature slots must hold exact zeros (they feed MFMA rows/cols >= p, which are ignored,
    // but must not carry NaN into the VALU side sums of lanes f >= p either)
    {
        typename Tile<T>::vec z;
#pragma unroll
        for (int e = 0; e < RPL; ++e) z[e] = T(0);
        for (int c = p; c < 16; ++c) *reinterpret_cast<typename Tile<T>::vec*>(wl + c * kColStride + lane * 16) = z;
    }

    // LEVM (WM == 4): the leverages on the MATRIX CORES.  h_i = z_i' P z_i with P = (X'X)^-1 =
    // x_i' (P_xx x_i + 2 p_xb) + p_bb (the intercept's 1 handled in closed form, so 16 features + intercept need no 17th operand row):
    // U = P_xx X' for 16 rows of the tile is FOUR matrix instructions (contraction over the 16 features, four at a time) whose result
    // layout -- lane (row j, quad q), register v holds U[feature drow(lane, v)][row j] -- is exactly the layout of their own B
    // operands, so the lane multiplies U by the x values it has just fed in, adds 2 p_xb, and the four quads are summed by two lane
    // swaps.  32 more matrix instructions per 128-row tile (as many as the Gram itself) replace 153 FMAs + 306 v_readlane PER ROW:
    // the HC2 / HC3 pass was bound by that vector work (report_c2_hc3 0.55 of the HBM peak against 0.73 for SE / HC1).
    static_assert(!(WM == 4 && P2 != 0), "HC2 / HC3 run on the unpacked tile (the packed kernels' vector form of the leverages -- 153 FMAs + 306 v_readlane per row, 744 spilled SGPRs -- lost to it at every width and left the product in round 6)");
    constexpr bool LEVM = WM == 4;
    T lev_a[4] = {T(0), T(0), T(0), T(0)}, lev_pb[4] = {T(0), T(0), T(0), T(0)};
    T lev_pbb = T(0);
    if constexpr (LEVM) {
        const T* const inv_g = static_cast<const T*>(ia.inv);
        const int pp = p + bias, i = lane & 15;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int fm = Tile<T>::drow(lane, m);  // the feature this lane's quad carries in matrix step m (operand A and B alike)
            lev_a[m] = (i < p && fm < p) ? inv_g[i + fm * pp] : T(0);
            lev_pb[m] = (bias && fm < p) ? T(2) * inv_g[fm + p * pp] : T(0);
        }
        lev_pbb = bias ? inv_g[p + p * pp] : T(0);
    }
    WaveAcc acc;
    zero_acc(acc);
    double hc_sw = 0.0, hc_ys = 0.0;  // HC passes on the unpacked tile: sum w / sum w y' of the lane's OWN rows, added to the record at the end
    double hy1 = 0.0, hy2 = 0.0;      // ia.y_sums: sum (y - y[0]), sum (y - y[0])^2 of the lane's own rows (the report's derived var(y))
    const int64_t nfull = n / TR;
    const int64_t wid = (int64_t)blockIdx.x * kWaves + wave, nw = (int64_t)gridDim.x * kWaves;
    TileRegs<T> regs;
    ColPtrs<T> cp;
    fetch_col_ptrs<T, LOADW>(cols, p, cp);
    const double yc = (ia.y_sums && n > 0) ? (double)cp.y[0] : 0.0;
    // WM == 2: coefficients are wave uniform (scalar registers); rows at or beyond n_lim get weight 0
    T bx[16];
    T b0 = T(0);
    if constexpr (WM >= 2) {
        if (WM == 2 || WM == 4 || !ia.init) {
#pragma unroll
            for (int c = 0; c < 16; ++c) bx[c] = (c < p) ? beta[c] : T(0);
            b0 = bias ? beta[p] : T(0);
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) bx[c] = T(0);
        }
    }
    auto resid_weights = [&](int64_t row, int64_t n_lim) __attribute__((always_inline)) {
        if constexpr (WM == 2) {
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
                T acc1 = b0;  // same order as pass2_kernel: bias first, then the features in column order
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < p) acc1 += regs.x[c][e] * bx[c];
                const T r = regs.y[e] - acc1;
                regs.w[e] = (row + e < n_lim) ? r * r : T(0);
                if constexpr (P2 == 0) hc_sw += (double)regs.w[e];  // (the unpacked tile's consume leaves sum w to the lane that owns the row)
                if (ia.y_sums) {
                    const double dy = (row + e < n_lim) ? (double)regs.y[e] - yc : 0.0;
                    hy1 += dy;
                    hy2 = fma(dy, dy, hy2);
                }
            }
        }
        if constexpr (WM == 3) {
            const T ymean = (T)ia.y_mean;
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
                const T yv = regs.y[e];
                T eta, mu;
                if (ia.init) {
                    mu = (ia.variance == 2) ? (yv + T(0.5)) * T(0.5) : (yv + ymean) * T(0.5);
                    eta = glm_link<T>(ia.link, mu);
                } else {
                    T acc1 = b0;
#pragma unroll
                    for (int c = 0; c < 16; ++c)
                        if (c < p) acc1 += regs.x[c][e] * bx[c];
                    eta = acc1;
                    mu = glm_inv<T>(ia.link, eta);
                }
                const T d = glm_deriv<T>(ia.link, mu);
                const T wv = T(1) / (d * d * glm_var<T>(ia.variance, mu));
                const T z = eta + d * (yv - mu);
                const bool in = row + e < n_lim;
                regs.w[e] = in ? wv : T(0);
                regs.y[e] = in ? z : T(0);
            }
        }
    };
    // Wave w streams the CONTIGUOUS tile range [nfull w / W, nfull (w+1) / W) of every column (consecutive 1 KiB pieces of a
    // column stay with one wave: 6.1 TB/s for the bare access pattern against 5.5 TB/s grid-strided, tools/membw.hip).
    // -DPDS_GRAM_STRIDED keeps the grid-strided walk (A/B).
#ifdef PDS_GRAM_STRIDED
    int64_t t = wid;
    const int64_t t_end = nfull, t_step = nw;
#else
    int64_t t = (int64_t)(((__int128)nfull * wid) / nw);
    const int64_t t_end = (int64_t)(((__int128)nfull * (wid + 1)) / nw), t_step = 1;
#endif
    // LEVM, first half (while the tile is still in registers): squared residuals of the lane's rows, the features to LDS
    T lev_r2[RPL];
    auto lev_stage = [&](int64_t row, int64_t n_lim) __attribute__((always_inline)) {
        using V = typename Tile<T>::vec;
#pragma unroll
        for (int e = 0; e < RPL; ++e) {
            T acc1 = b0;  // same order as pass2_kernel: bias first, then the features in column order
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < p) acc1 += regs.x[c][e] * bx[c];
            const T r = regs.y[e] - acc1;
            lev_r2[e] = r * r;
            if (ia.y_sums) {
                const double dy = (row + e < n_lim) ? (double)regs.y[e] - yc : 0.0;
                hy1 += dy;
                hy2 = fma(dy, dy, hy2);
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < p) *reinterpret_cast<V*>(wl + c * kColStride + lane * 16) = regs.x[c];
    };
    // LEVM, second half (the next tile may already be on its way into `regs`): leverages of the tile's rows from LDS on the matrix
    // cores, then the weight column w = e^2 / (1 - h)^k and the y slot (1 - h)^k of the lane's own rows
    auto lev_weights = [&](int64_t row, int64_t n_lim) __attribute__((always_inline)) {
        using V = typename Tile<T>::vec;
        using Acc = typename Tile<T>::acc;
        const int j = lane & 15;
        T* const hcol = reinterpret_cast<T*>(wl + kSlotW * kColStride);
        const T* xq[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xq[m] = reinterpret_cast<const T*>(wl + Tile<T>::drow(lane, m) * kColStride) + j;
        // four 16-row blocks at a time, their matrix steps interleaved (block-inner order): consecutive instructions are independent,
        // a block's own chain comes round every fourth issue -- left to the compiler the blocks ran one after the other, every step
        // waiting out the previous one's 16 passes (s_nop 15 in front of each reduction: ~400 clk per block with the pipe idle)
        constexpr int NB = PDS_LEV_NB;
#pragma unroll
        for (int b0 = 0; b0 < TR / 16; b0 += NB) {
            T xb[NB][4];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb)
#pragma unroll
                for (int m = 0; m < 4; ++m) xb[bb][m] = xq[m][16 * (b0 + bb)];
            Acc u[NB];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {  // (the accumulator starts at 2 p_xb in its own layout: the matrix steps add P_xx x to it for nothing)
                if constexpr (sizeof(T) == 8) u[bb] = Acc{lev_pb[0], lev_pb[1], lev_pb[2], lev_pb[3]};
                else u[bb] = Acc{0, 0, 0, 0};
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                // (f64 operand layout: step m carries features 4 m .. 4 m + 3 -- narrower frames stop early; f32: every step
                //  carries features m, 4 + m, 8 + m, 12 + m)
                if (sizeof(T) == 8 && 4 * m >= p) continue;
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) u[bb] = Tile<T>::mfma(lev_a[m], xb[bb][m], u[bb]);
            }
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                T hl = T(0);
#pragma unroll
                for (int m = 0; m < 4; ++m) hl = fma(xb[bb][m], sizeof(T) == 8 ? (T)u[bb][m] : (T)u[bb][m] + lev_pb[m], hl);
                if constexpr (sizeof(T) == 8) {
                    hl = xor_sum_q(hl);
                } else {
                    hl += __shfl_xor(hl, 16);
                    hl += __shfl_xor(hl, 32);
                }
                if (lane < 16) hcol[16 * (b0 + bb) + j] = hl + lev_pbb;
            }
        }
        const V hv = *reinterpret_cast<const V*>(wl + kSlotW * kColStride + lane * 16);
        V wv, yv;
#pragma unroll
        for (int e = 0; e < RPL; ++e) {
            const T om = T(1) - hv[e];
            const T s2 = lev_r2[e];
            const T wgt = (ia.hc_pow == 1) ? s2 * (T(1) / om) : s2 * (T(1) / (om * om));
            const bool in = row + e < n_lim;
            wv[e] = in ? wgt : T(0);
            yv[e] = in ? ((ia.hc_pow == 1) ? om : om * om) : T(0);
            hc_sw += (double)wv[e];
            hc_ys = fma((double)wv[e], (double)yv[e], hc_ys);
        }
        *reinterpret_cast<V*>(wl + kSlotW * kColStride + lane * 16) = wv;
        *reinterpret_cast<V*>(wl + kSlotY * kColStride + lane * 16) = yv;
    };
    if (t < t_end) load_full_tile<T, LOADW>(cp, p, t * TR + lane * RPL, regs);
    for (; t < t_end; t += t_step) {
        const int64_t tn = t + t_step;
        if constexpr (LEVM) {
            lev_stage(0, 1 << 30);
            if (tn < t_end) load_full_tile<T, LOADW>(cp, p, tn * TR + lane * RPL, regs);
            lev_weights(0, 1 << 30);
        } else {
            resid_weights(0, 1 << 30);  // (full tiles: every row counts)
            store_tile_lds<T, WEIGHTED>(wl, p, lane, regs);
            if (tn < t_end) load_full_tile<T, LOADW>(cp, p, tn * TR + lane * RPL, regs);
        }
        if constexpr (P2 != 0) consume_tile_pack<T, WEIGHTED, P2>(wl, lane, acc);
        else consume_tile<T, WEIGHTED, (WM == 2 || WM == 4) ? 2 : 0>(wl, lane, TR / 4, acc);  // (HC passes: a meat block; sum w / sum w y' per tile below)
    }
    if (nfull * TR < n && wid == nw - 1) {  // ragged tail: exactly one wave
        load_tail_tile<T, LOADW>(cp, p, nfull * TR + lane * RPL, n, regs);
        if constexpr (LEVM) {
            lev_stage(nfull * TR + lane * RPL, n);
            lev_weights(nfull * TR + lane * RPL, n);
        } else {
            resid_weights(nfull * TR + lane * RPL, n);
            store_tile_lds<T, WEIGHTED>(wl, p, lane, regs);
        }
        if constexpr (P2 != 0) consume_tile_pack<T, WEIGHTED, P2>(wl, lane, acc);
        else consume_tile<T, WEIGHTED, (WM == 2 || WM == 4) ? 2 : 0>(wl, lane, TR / 4, acc);  // (HC passes: a meat block; sum w / sum w y' per tile below)
    }

    if constexpr ((WM == 2 || WM == 4) && P2 == 0) {
        // the per-lane sums of the lane's own rows -> the record's convention (lane 0 of the four row-slot lanes carries the wave's total)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            hc_sw += __shfl_xor(hc_sw, o);
            hc_ys += __shfl_xor(hc_ys, o);
        }
        acc.sw = lane == 0 ? hc_sw : 0.0;
        acc.ys = lane == 0 ? hc_ys : 0.0;
    }
    // block reduction through LDS (tile storage is dead now)
    __syncthreads();
    double* recs = reinterpret_cast<double*>(smem);
    if constexpr (P2 != 0) wave_record_pack<T, P2>(acc, lane, recs + wave * kPartStride);
    else wave_record<T>(acc, lane, recs + wave * kPartStride);
    {   // the two spare slots behind sum w: the wave's sums of y - y[0] and its square (zeros unless ia.y_sums)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            hy1 += __shfl_xor(hy1, o);
            hy2 += __shfl_xor(hy2, o);
        }
        if (lane == 0) {
            recs[wave * kPartStride + kPartY1] = hy1;
            recs[wave * kPartStride + kPartY2] = hy2;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kPartStride; e += blockDim.x) {
        double s = 0.0;
        if (e <= kPartY2) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s += recs[w * kPartStride + e];
        }
        partials[(int64_t)blockIdx.x * kPartStride + e] = s;
    }
}

// =============================================================================================
// Row-major frames (the pyclass route: a NumPy / torch matrix X[n][ld], y[n]; numpy_faer.rs:10-66 reads it through a
// strided MatRef).  Row-major IS the matrix core's operand layout: lane (f = lane & 15, q = lane >> 4) of a 16x16x4 step
// holds feature f of row 4 s + q, i.e. the 64 lanes of one load instruction read four consecutive rows of 16 features
// -- 512 contiguous bytes at ld = 16 -- and the value goes into the MFMA as it arrives: no LDS tile, no transposition pass.
// Each wave streams a contiguous row range, eight 4-row steps (32 rows) per iteration with the next iteration's sixteen
// loads in flight in a second register set.  Same partial records and finalize kernel as moments_small_kernel.
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256, 2) void moments_rowmajor_kernel(const T* __restrict__ X, int64_t ld, const T* __restrict__ y,
                                                                  int p, int64_t n, double* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int U = 8;  // 4-row steps per iteration
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = lane & 15, q = lane >> 4;
    const bool feat = f < p;
    const gptr<T> Xg = as_global(X);
    const gptr<T> yg = as_global(y);
    const int64_t wid = (int64_t)blockIdx.x * kWaves + wave, nw = (int64_t)gridDim.x * kWaves;
    const int64_t nit = (n + 4 * U - 1) / (4 * U);  // iterations of 32 rows; the last one may be ragged
    int64_t it = (int64_t)(((__int128)nit * wid) / nw);
    const int64_t it_end = (int64_t)(((__int128)nit * (wid + 1)) / nw);
    using Acc = typename Tile<T>::acc;
    WaveAcc acc;
    zero_acc(acc);
    T xa[U], ya[U];
    auto load_iter = [&](int64_t i, T (&xv)[U], T (&yv)[U]) __attribute__((always_inline)) {
        const int64_t r0 = i * (4 * U) + q;
        if (r0 + 4 * (U - 1) < n - 3 + q) {  // every row of the iteration exists (rows r0 + 4 u <= n - 1 for all q)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = r0 + 4 * u;
                xv[u] = feat ? __builtin_nontemporal_load(Xg + r * ld + f) : T(0);
                yv[u] = __builtin_nontemporal_load(yg + r);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = r0 + 4 * u;
                const bool in = r < n;
                xv[u] = (in && feat) ? Xg[r * ld + f] : T(0);
                yv[u] = in ? yg[r] : T(0);
            }
        }
    };
    if (it < it_end) load_iter(it, xa, ya);
    for (; it < it_end; ++it) {
        T xb[U], yb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xb[u] = xa[u];
            yb[u] = ya[u];
        }
        if (it + 1 < it_end) load_iter(it + 1, xa, ya);
        if constexpr (sizeof(T) == 8) {
            Acc a = Acc{acc.d[0], acc.d[1], acc.d[2], acc.d[3]};
            double xy = acc.xy, cs = acc.cs, yy = acc.yy, ys = acc.ys;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a = Tile<T>::mfma(xb[u], xb[u], a);
                xy = fma(xb[u], yb[u], xy);
                cs += xb[u];
                yy = fma(yb[u], yb[u], yy);
                ys += yb[u];
            }
            acc.d[0] = a[0]; acc.d[1] = a[1]; acc.d[2] = a[2]; acc.d[3] = a[3];
            acc.xy = xy; acc.cs = cs; acc.yy = yy; acc.ys = ys;
        } else {  // f32: one iteration's products in f32 on the matrix core, folded into f64 accumulators (like the tile kernel)
            Acc a = Acc{0, 0, 0, 0};
            float xy = 0.f, cs = 0.f, yy = 0.f, ys = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a = Tile<T>::mfma(xb[u], xb[u], a);
                xy = fmaf(xb[u], yb[u], xy);
                cs += xb[u];
                yy = fmaf(yb[u], yb[u], yy);
                ys += yb[u];
            }
            acc.d[0] += (double)a[0]; acc.d[1] += (double)a[1]; acc.d[2] += (double)a[2]; acc.d[3] += (double)a[3];
            acc.xy += (double)xy; acc.cs += (double)cs; acc.yy += (double)yy; acc.ys += (double)ys;
        }
    }
    __syncthreads();
    double* recs = reinterpret_cast<double*>(smem);
    wave_record<T>(acc, lane, recs + wave * kPartStride);
    __syncthreads();
    for (int e = threadIdx.x; e < kPartStride; e += blockDim.x) {
        double sum = 0.0;
        if (e <= kPartSW) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) sum += recs[w * kPartStride + e];
        }
        partials[(int64_t)blockIdx.x * kPartStride + e] = sum;
    }
}

// the blocks' y sums (slots kPartY1 / kPartY2 of the partial records) in block order -> out[0..1]
__global__ __launch_bounds__(64) void moments_ysums_kernel(const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 64) {
        a += partials[(int64_t)i * kPartStride + kPartY1];
        b += partials[(int64_t)i * kPartStride + kPartY2];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    if (threadIdx.x == 0) {
        out[0] = a;
        out[1] = b;
    }
}

// fixed-order reduction of the per-block partials and assembly of the (p+2)x(p+2) moment matrix.
// One 64-lane block per record element: lane l sums blocks l, l+64, ... in order, then a fixed
// butterfly combines the lanes -- deterministic, and ~2 us instead of a 512-long dependent load chain.
template <typename T>
__global__ __launch_bounds__(64) void moments_finalize_kernel(const double* __restrict__ partials, int nblocks,
                                                              int p, double n_rows, int weighted, int p2,
                                                              T* __restrict__ out, double* __restrict__ sums_out,
                                                              int sums_from_ys = 0) {
    const int e = blockIdx.x;  // element of the partial record, 0 .. kPartSW
    auto blocksum = [&](int el) {
        double v = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += 64) v += partials[(int64_t)b * kPartStride + el];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        return v;
    };
    // packed build (p2 = feature slots per slab): fold the slabs -- diagonal p2 x p2 blocks of the tile, xy / cs lanes f % p2
    const int nsl = p2 ? 16 / p2 : 1, st = p2 ? p2 : 0;
    double s = 0.0;
    if (e < kPartXY) {
        const int i = e & 15, j = e >> 4;
        if (p2 && (i >= p2 || j >= p2)) return;
        for (int k = 0; k < nsl; ++k) s += blocksum((k * st + i) + 16 * (k * st + j));
    } else if (e < kPartYY) {  // xy and cs lanes
        const int base = e < kPartCS ? kPartXY : kPartCS, i = e - base;
        if (p2 && i >= p2) return;
        for (int k = 0; k < nsl; ++k) s += blocksum(base + k * st + i);
    } else {
        s = blocksum(e);
    }
    if (threadIdx.x != 0) return;
    const int q = p + 2;
    if (e < kPartXY) {  // D[i + 16 j]: take the upper triangle and mirror it
        const int i = e & 15, j = e >> 4;
        if (i <= j && j < p) {
            out[i + j * q] = (T)s;
            out[j + i * q] = (T)s;
        }
    } else if (e < kPartCS) {
        const int i = e - kPartXY;
        if (i < p) {
            out[i + (p + 1) * q] = (T)s;
            out[(p + 1) + i * q] = (T)s;
        }
    } else if (e < kPartYY) {
        const int i = e - kPartCS;
        if (i < p) {
            out[i + p * q] = (T)s;
            out[p + i * q] = (T)s;
        }
    } else if (e == kPartYY) {
        out[(p + 1) + (p + 1) * q] = (T)s;
    } else if (e == kPartYS) {
        out[p + (p + 1) * q] = (T)s;
        out[(p + 1) + p * q] = (T)s;
        if (sums_out && sums_from_ys) {  // HC2 / HC3: the y slot carried e^2 / w, so sum(w y') is the residual sum of squares
            sums_out[0] = s;
            sums_out[1] = 0.0;
        }
    } else if (e == kPartSW) {
        out[p + p * q] = (T)(weighted ? s : n_rows);
        if (sums_out && !sums_from_ys) {  // residual-weighted build: sum(w) IS the residual sum of squares (report_second_pass reads it here)
            sums_out[0] = s;
            sums_out[1] = 0.0;
        }
    }
}

// =============================================================================================
// grouped: one wave per group at a time (rows of a group are contiguous, group starts are only
// element-aligned -> 8/4-byte lane loads, still 512/256 B coalesced per instruction)
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256, 2) void grouped_moments_kernel(const T* const* __restrict__ cols, int p,
                                                                 const int64_t* __restrict__ offsets,
                                                                 int64_t n_groups, T* __restrict__ out,
                                                                 const int32_t* __restrict__ gidx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* wl = smem + wave * kWaveLds;
    double* rec = reinterpret_cast<double*>(wl);  // aliases the (dead) tile between two groups
    const int q = p + 2;
    const int64_t wid = (int64_t)blockIdx.x * kWaves + wave, nw = (int64_t)gridDim.x * kWaves;
    GroupRegs<T> regs;
    ColPtrs<T> cp;
    fetch_col_ptrs<T, false>(cols, p, cp);
    int64_t g = wid;
    int64_t r0 = 0, rend = 0;
    // gidx != null: record g is the Gram matrix of group gidx[g] (the pivoted-QR pass over the groups the fused kernel
    // marked: a sparse subset of the frame's groups, records compact)
    auto bounds = [&](int64_t k) __attribute__((always_inline)) {
        const int64_t gg = gidx ? (int64_t)gidx[k] : k;
        r0 = offsets[gg];
        rend = offsets[gg + 1];
    };
    if (g < n_groups) {
        bounds(g);
        load_group_tile<T>(cp, p, r0, rend, lane, regs);
    }
    for (; g < n_groups; g += nw) {
        WaveAcc acc;
        zero_acc(acc);
        const int64_t gr0 = r0, grend = rend;
        // first 128 rows of this group are already in registers
        store_group_lds<T>(wl, p, lane, regs);
        int64_t done = 128;
        const int64_t ng = grend - gr0;
        const int64_t gn = g + nw;
        if (ng <= 128 && gn < n_groups) {  // prefetch the next group's first tile
            bounds(gn);
            load_group_tile<T>(cp, p, r0, rend, lane, regs);
        }
        {
            const int rows = (int)(ng < 128 ? ng : 128);
            consume_tile<T, false>(wl, lane, (rows + 3) >> 2, acc);
        }
        while (done < ng) {  // long groups: stream the rest 128 rows at a time
            load_group_tile<T>(cp, p, gr0 + done, grend, lane, regs);
            store_group_lds<T>(wl, p, lane, regs);
            const int64_t left = ng - done;
            const int rows = (int)(left < 128 ? left : 128);
            consume_tile<T, false>(wl, lane, (rows + 3) >> 2, acc);
            done += 128;
            if (done >= ng && gn < n_groups) {
                bounds(gn);
                load_group_tile<T>(cp, p, r0, rend, lane, regs);
            }
        }
        // assemble A for this group
        wave_record<T>(acc, lane, rec);
        T* o = out + g * (int64_t)(q * q);
        for (int idx = lane; idx < q * q; idx += 64) {
            int i = idx % q, j = idx / q;
            if (i > j) { int tmp = i; i = j; j = tmp; }
            double v;
            if (j < p) v = rec[kPartD + i + 16 * j];
            else if (j == p) v = (i < p) ? rec[kPartCS + i] : (double)ng;
            else v = (i < p) ? rec[kPartXY + i] : (i == p ? rec[kPartYS] : rec[kPartYY]);
            o[idx] = (T)v;
        }
    }
}

// =============================================================================================
// host launchers
// =============================================================================================
template <typename T>
int launch_moments(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, bool weighted,
                   T* d_moments, const T* d_beta_resid, int bias_resid, double* d_sums_resid, double* d_moments_f64,
                   const IrlsArgs* irls, double* d_ysums) {
    if (n_feat > kMaxFeatSmall) {
        if (d_moments_f64) return fail(PDS_ERR_INVALID, "internal: f64 moment slots are the p <= 16 kernel's");
        if (d_beta_resid) return fail(PDS_ERR_INVALID, "internal: the residual-weighted Gram build is the p <= 16 kernel's");
        return launch_moments_wide<T>(ctx, dc, n_feat, n_rows, weighted, d_moments);
    }
    if ((d_beta_resid || irls) && weighted) return fail(PDS_ERR_INVALID, "internal: residual / IRLS weights replace the weight column");
    if (irls && n_feat > kMaxFeatSmall) return fail(PDS_ERR_UNSUPPORTED, "GLM (IRLS): up to 16 feature columns");
    IrlsArgs ia = irls ? *irls : IrlsArgs{};
    if (d_ysums && d_beta_resid && (!irls || irls->hc_pow > 0)) ia.y_sums = 1;  // (the residual-weighted passes of the report)
    else d_ysums = nullptr;
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    constexpr int TR = 64 * Tile<T>::RPL;
    int64_t ntiles = (n_rows + TR - 1) / TR;
    int64_t want = (ntiles + kWaves - 1) / kWaves;
    int nblocks = (int)std::min<int64_t>(std::max<int64_t>(want, 1), (int64_t)ctx->num_cus * 2);
    double* partials = ctx->partials;  // sized for 8 blocks per CU at context creation
    KernelTimer timer(ctx, kKindMoments);
    size_t lds = (size_t)kWaves * kWaveLds + 64;  // + slack: the pipelined operand fetch reads two steps ahead
    const bool hc23 = irls && irls->hc_pow > 0;  // residual weights scaled by the leverages (HC2 / HC3)
    // (HC2 / HC3 at any width: the unpacked tile, whose leverages run on the matrix cores -- the packed kernels' vector form of the
    //  leverages cost more than the packing saves: 1e8 x 8, hc3 5.2 ms against 4.8 at 9 features)
    const int p2 = (n_feat > 8 || hc23) ? 0 : (n_feat > 4 ? 8 : (n_feat > 2 ? 4 : (n_feat > 1 ? 2 : 1)));
    auto launch = [&](auto w_c, auto p16_c, auto p2_c) {
        hipLaunchKernelGGL((moments_small_kernel<T, decltype(w_c)::value, decltype(p16_c)::value, decltype(p2_c)::value>), dim3(nblocks),
                           dim3(256), lds, ctx->stream, dc.d_ptrs, n_feat, n_rows, partials, d_beta_resid, bias_resid, ia);
    };
    auto by_p2 = [&](auto w_c) {
        using std::integral_constant;
        using std::false_type;
        if (n_feat == 16) launch(w_c, std::true_type{}, integral_constant<int, 0>{});
        else if constexpr (decltype(w_c)::value == 4) launch(w_c, false_type{}, integral_constant<int, 0>{});  // (p2 = 0 above: HC2 / HC3 never pack)
        else if (p2 == 8) launch(w_c, false_type{}, integral_constant<int, 8>{});
        else if (p2 == 4) launch(w_c, false_type{}, integral_constant<int, 4>{});
        else if (p2 == 2) launch(w_c, false_type{}, integral_constant<int, 2>{});
        else if (p2 == 1) launch(w_c, false_type{}, integral_constant<int, 1>{});
        else launch(w_c, false_type{}, integral_constant<int, 0>{});
    };
    if (hc23) by_p2(std::integral_constant<int, 4>{});
    else if (irls) by_p2(std::integral_constant<int, 3>{});
    else if (d_beta_resid) by_p2(std::integral_constant<int, 2>{});
    else if (weighted) by_p2(std::integral_constant<int, 1>{});
    else by_p2(std::integral_constant<int, 0>{});
    if (d_moments_f64)  // one row chunk of a host frame: the chunk's record stays in f64 until the chunks are summed
        hipLaunchKernelGGL((moments_finalize_kernel<double>), dim3(kPartSW + 1), dim3(64), 0, ctx->stream, partials, nblocks, n_feat,
                           (double)n_rows, (weighted || d_beta_resid || irls) ? 1 : 0, p2, d_moments_f64, (double*)nullptr, 0);
    else
        hipLaunchKernelGGL((moments_finalize_kernel<T>), dim3(kPartSW + 1), dim3(64), 0, ctx->stream, partials, nblocks, n_feat,
                           (double)n_rows, (weighted || d_beta_resid || irls) ? 1 : 0, p2, d_moments,
                           (d_beta_resid && (!irls || hc23)) ? d_sums_resid : nullptr, hc23 ? 1 : 0);
    if (d_ysums) hipLaunchKernelGGL(moments_ysums_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double*)partials, nblocks, d_ysums);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}


// ---------------------------------------------------------------------------------------------
// grouped Gram build for 17 .. 64 features: one wave per group, Z = [x | 1 | y] (q = p + 2 <= 66 columns) as
// NB = ceil(q / 16) blocks of 16 columns.  32 rows at a time are staged column-major in LDS (f32 frames are widened to
// f64 on the way in), every 4-row step feeds the NB operand registers to the upper-triangular block pairs of
// v_mfma_f64_16x16x4_f64 (<= 15 accumulator tiles = 120 VGPRs), and the finished group writes its full symmetric
// (p+2)^2 moment record.  Two-kernel pipeline (record -> solve_wave.hip's register solver): at these widths a group's record is as
// large as its rows, so nothing is gained by fusing.  f64 frames with 28 .. 64 features take the streamed form instead
// (grouped_mid.hip, grouped_mid_stream_kernel); this kernel serves 17 .. 27 features and f32 frames.
// ---------------------------------------------------------------------------------------------
constexpr int kMidRows = 32;
constexpr int kMidStride = 34;  // doubles per LDS column: 68 dwords = 4 mod 64 -> conflict-free b64 operand reads

template <typename T, int NB>
__global__ __launch_bounds__(64) void grouped_moments_mid_kernel(const T* const* __restrict__ cols, int p,
                                                                 const int64_t* __restrict__ offsets, int64_t n_groups,
                                                                 T* __restrict__ moments) {
    constexpr int NT = NB * (NB + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) double tile[];  // [NB * 16][kMidStride]
    const int lane = threadIdx.x;
    const int q = p + 2;
    const int f = lane & 15, kq = lane >> 4;
    for (int i = lane; i < NB * 16 * kMidStride; i += 64) tile[i] = 0.0;  // columns >= q stay zero for good
    PDS_WAVE_LDS_SYNC();
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t r0 = offsets[g], r1 = offsets[g + 1];
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
        for (int64_t base = r0; base < r1; base += kMidRows) {
            // stage 32 rows: lanes 0-31 take an even column, lanes 32-63 the next one
            const int row = lane & 31;
            const int64_t r = base + row;
            const bool in = r < r1;
            // all loads of the stage first (unconditional, clamped column and row: a load under its own exec mask is serialised
            // behind the previous one by the compiler's vmcnt(0) -- 33 HBM round trips per 32 rows made this kernel run at
            // 0.1 - 0.3 TB/s), then the LDS stores
            const int64_t rc = in ? r : r1 - 1;
            T vreg[NB * 8];
#pragma unroll
            for (int k = 0; k < NB * 8; ++k) {
                const int c = 2 * k + (lane >> 5);
                const int cc = c < p ? c : p;  // (column p of the table is y: it serves c == p + 1, the ones column and the padding)
                vreg[k] = as_global(cols[cc])[rc];
            }
#pragma unroll
            for (int k = 0; k < NB * 8; ++k) {
                const int c = 2 * k + (lane >> 5);
                if (c < q) tile[c * kMidStride + row] = !in ? 0.0 : (c == p ? 1.0 : (double)vreg[k]);
            }
            PDS_WAVE_LDS_SYNC();
#pragma unroll
            for (int s = 0; s < kMidRows / 4; ++s) {
                double a[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) a[b] = tile[(16 * b + f) * kMidStride + 4 * s + kq];
                int t = 0;
#pragma unroll
                for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                    for (int bj = bi; bj < NB; ++bj) {
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bi], a[bj], acc[t], 0, 0, 0);
                        ++t;
                    }
            }
            PDS_WAVE_LDS_SYNC();
        }
        // ---- record: D tile (bi, bj) holds rows 16 bi + (lane >> 4) + 4 r, column 16 bj + (lane & 15)
        T* M = moments + g * (int64_t)q * q;
        int t = 0;
#pragma unroll
        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int bj = bi; bj < NB; ++bj) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int i = 16 * bi + kq + 4 * rr, j = 16 * bj + f;
                    if (i < q && j < q) {
                        const T v = (T)acc[t][rr];
                        M[i + (int64_t)j * q] = v;
                        if (bi != bj) M[j + (int64_t)i * q] = v;
                    }
                }
                ++t;
            }
    }
}

template <typename T, int NB>
static void launch_mid_nb(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, const int64_t* d_offsets, int64_t n_groups,
                          T* d_moments) {
    const size_t lds = (size_t)NB * 16 * kMidStride * sizeof(double);
    const int per_cu = std::max(1, std::min(8, (int)((160 * 1024) / lds)));
    const int nblocks = (int)std::min<int64_t>(n_groups, (int64_t)ctx->num_cus * per_cu);
    hipLaunchKernelGGL((grouped_moments_mid_kernel<T, NB>), dim3(nblocks), dim3(64), lds, ctx->stream, dc.d_ptrs, n_feat,
                       d_offsets, n_groups, d_moments);
}

template <typename T>
int launch_grouped_moments(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, const int64_t* d_offsets,
                           int64_t n_groups, T* d_moments, const int32_t* d_group_index) {
    if (n_feat < 1 || n_feat > kMaxFeatWide)
        return fail(PDS_ERR_UNSUPPORTED, "grouped regressions: 1..64 features supported");
    if (n_groups <= 0) return PDS_OK;
    if (d_group_index && n_feat > kMaxFeatSmall) return fail(PDS_ERR_INVALID, "internal: indexed grouped Gram build is the p <= 16 kernel's");
    if (n_feat > kMaxFeatSmall) {
        const int nb = (n_feat + 2 + 15) / 16;  // 2 .. 5
        KernelTimer timer(ctx, kKindGroupedMoments);
        switch (nb) {
            case 2: launch_mid_nb<T, 2>(ctx, dc, n_feat, d_offsets, n_groups, d_moments); break;
            case 3: launch_mid_nb<T, 3>(ctx, dc, n_feat, d_offsets, n_groups, d_moments); break;
            case 4: launch_mid_nb<T, 4>(ctx, dc, n_feat, d_offsets, n_groups, d_moments); break;
            default: launch_mid_nb<T, 5>(ctx, dc, n_feat, d_offsets, n_groups, d_moments); break;
        }
        PDS_HIP_CHECK(hipGetLastError());
        return PDS_OK;
    }
    int64_t want = (n_groups + kWaves - 1) / kWaves;
    int nblocks = (int)std::min<int64_t>(want, (int64_t)ctx->num_cus * 2);
    size_t lds = (size_t)kWaves * kWaveLds;
    KernelTimer timer(ctx, kKindGroupedMoments);
    hipLaunchKernelGGL((grouped_moments_kernel<T>), dim3(nblocks), dim3(256), lds, ctx->stream, dc.d_ptrs, n_feat,
                       d_offsets, n_groups, d_moments, d_group_index);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

// f64, 16 bytes per lane: lane L loads the feature PAIR (2 f', 2 f' + 1), f' = L & 7, of row slot L >> 3 -- the 64 lanes of
// one instruction read 8 consecutive rows of 16 features = 1 KiB contiguous at ld = 16, like the column kernels' loads.
// As an MFMA operand, lane (i = L & 15, k = L >> 4) then carries feature 2 (i & 7) [.x] / 2 (i & 7) + 1 [.y] of row
// 2 k + (i >> 3): three matrix instructions per 8 rows -- x.x, x.y, y.y -- whose two diagonal 8 x 8 blocks (rows 2 k and
// rows 2 k + 1) sum to the even-even, even-odd and odd-odd blocks of the Gram matrix (the off-diagonal blocks are cross terms
// of different rows nobody reads).  1.5x the matrix work of the column kernel (1.5 ms of pipe per 1e8 rows, still under the
// HBM time) for loads that are as wide as the link likes them: 8 B per lane ran at 0.39 of the HBM peak.
#ifndef PDS_ROWMAJOR_U
#define PDS_ROWMAJOR_U 3
#endif
#ifndef PDS_ROWMAJOR_BLOCKS
#define PDS_ROWMAJOR_BLOCKS 4
#endif
__global__ __launch_bounds__(256, PDS_ROWMAJOR_BLOCKS) void moments_rowmajor_f64_kernel(const double* __restrict__ X, int64_t ld,
                                                                      const double* __restrict__ y, int p, int64_t n,
                                                                      double* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int U = PDS_ROWMAJOR_U;  // 8-row steps per iteration.  Measured at 1e8 x 16 (profiles/r02_rowmajor_ab.txt): four resident blocks (16
    // waves per CU) with U = 2 / 3 / 4 / 6: 2.93 / 2.64 / 2.83 / 3.51 ms; U = 8 x 3 blocks 3.24, U = 16 x 2 blocks 4.89: many waves with short bursts beat
    // few waves with deep ones here (the y loads are 8-fold redundant per instruction, the matrix pipe runs 1.5x the column kernel's work)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fp = lane & 7, rs = lane >> 3;
    const int f0 = 2 * fp;
    const bool has0 = f0 < p, has1 = f0 + 1 < p;
    const gptr<double> Xg = as_global(X);
    const gptr<double> yg = as_global(y);
    const int64_t wid = (int64_t)blockIdx.x * kWaves + wave, nw = (int64_t)gridDim.x * kWaves;
    const int64_t nit = (n + 8 * U - 1) / (8 * U);
    int64_t it = (int64_t)(((__int128)nit * wid) / nw);
    const int64_t it_end = (int64_t)(((__int128)nit * (wid + 1)) / nw);
    d4 ee = {0, 0, 0, 0}, eo = {0, 0, 0, 0}, oo = {0, 0, 0, 0};
    double xy0 = 0, xy1 = 0, cs0 = 0, cs1 = 0, yy = 0, ys = 0;
    d2u xa[U];
    double ya[U];
    auto load_iter = [&](int64_t i, d2u (&xv)[U], double (&yv)[U]) __attribute__((always_inline)) {
        const int64_t r0 = i * (8 * U) + rs;
        const bool full = (i + 1) * (8 * U) <= n;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + 8 * u;
            const bool in = full || r < n;
            d2u v = {0.0, 0.0};
            if (in && has1) v = __builtin_nontemporal_load(reinterpret_cast<gptr<d2u>>(Xg + r * ld + f0));
            else if (in && has0) v[0] = Xg[r * ld + f0];
            xv[u] = v;
            yv[u] = in ? __builtin_nontemporal_load(yg + r) : 0.0;
        }
    };
    if (it < it_end) load_iter(it, xa, ya);
    for (; it < it_end; ++it) {
        d2u xb[U];
        double yb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xb[u] = xa[u];
            yb[u] = ya[u];
        }
        if (it + 1 < it_end) load_iter(it + 1, xa, ya);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double a = xb[u][0], b = xb[u][1], t = yb[u];
            ee = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, ee, 0, 0, 0);
            eo = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, eo, 0, 0, 0);
            oo = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, oo, 0, 0, 0);
            xy0 = fma(a, t, xy0);
            xy1 = fma(b, t, xy1);
            cs0 += a;
            cs1 += b;
            yy = fma(t, t, yy);
            ys += t;
        }
    }
    // ---- wave record in the layout of moments_small_kernel (D tile 16 x 16, xy, cs, yy, ys)
    __syncthreads();
    double* raw = reinterpret_cast<double*>(smem) + kWaves * kPartStride + wave * 768;  // three raw 16 x 16 product tiles
    double* rec = reinterpret_cast<double*>(smem) + wave * kPartStride;
    {
        const int j = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = (lane >> 4) + 4 * r;  // D layout of v_mfma_f64_16x16x4: col = lane & 15, row = (lane >> 4) + 4 reg
            raw[i + 16 * j] = ee[r];
            raw[256 + i + 16 * j] = eo[r];
            raw[512 + i + 16 * j] = oo[r];
        }
    }
    // per-feature sums: over the 8 row slots (lane bits 3..5)
#pragma unroll
    for (int o = 8; o <= 32; o <<= 1) {
        xy0 += __shfl_xor(xy0, o);
        xy1 += __shfl_xor(xy1, o);
        cs0 += __shfl_xor(cs0, o);
        cs1 += __shfl_xor(cs1, o);
        yy += __shfl_xor(yy, o);
        ys += __shfl_xor(ys, o);
    }
    PDS_WAVE_LDS_SYNC();
    for (int e = lane; e < 256; e += 64) {
        const int a = e & 15, b = e >> 4;  // features
        const int ia = a >> 1, ib = b >> 1;
        double v;
        if (!(a & 1) && !(b & 1)) v = raw[ia + 16 * ib] + raw[(ia + 8) + 16 * (ib + 8)];
        else if (!(a & 1) && (b & 1)) v = raw[256 + ia + 16 * ib] + raw[256 + (ia + 8) + 16 * (ib + 8)];
        else if ((a & 1) && !(b & 1)) v = raw[256 + ib + 16 * ia] + raw[256 + (ib + 8) + 16 * (ia + 8)];
        else v = raw[512 + ia + 16 * ib] + raw[512 + (ia + 8) + 16 * (ib + 8)];
        rec[kPartD + a + 16 * b] = v;
    }
    if (lane < 8) {
        rec[kPartXY + 2 * lane] = xy0;
        rec[kPartXY + 2 * lane + 1] = xy1;
        rec[kPartCS + 2 * lane] = cs0;
        rec[kPartCS + 2 * lane + 1] = cs1;
    }
    if (lane == 0) {
        rec[kPartYY] = yy;
        rec[kPartYS] = ys;
        rec[kPartSW] = 0.0;
    }
    __syncthreads();
    const double* recs = reinterpret_cast<const double*>(smem);
    for (int e = threadIdx.x; e < kPartStride; e += blockDim.x) {
        double sum = 0.0;
        if (e <= kPartSW) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) sum += recs[w * kPartStride + e];
        }
        partials[(int64_t)blockIdx.x * kPartStride + e] = sum;
    }
}

template <typename T>
int launch_moments_rowmajor(pds_ctx* ctx, const T* d_X, int64_t ld, const T* d_y, int n_feat, int64_t n_rows, T* d_moments,
                            double* d_moments_f64) {
    if (n_feat < 1 || n_feat > kMaxFeatSmall) return fail(PDS_ERR_INVALID, "internal: the row-major Gram kernel takes 1..16 features");
    const int64_t nit = (n_rows + 31) / 32;
    const int per_cu = sizeof(T) == 8 ? PDS_ROWMAJOR_BLOCKS : 2;
    const int nblocks = (int)std::min<int64_t>(std::max<int64_t>((nit + kWaves - 1) / kWaves, 1), (int64_t)ctx->num_cus * per_cu);
    double* partials = ctx->partials;
    KernelTimer timer(ctx, kKindMoments);
    if constexpr (sizeof(T) == 8) {
        static const bool narrow = [] { const char* e = dev_env("PDS_ROWMAJOR_NARROW"); return e && e[0] == '1'; }();  // (A/B)
        if (!narrow)
            hipLaunchKernelGGL(moments_rowmajor_f64_kernel, dim3(nblocks), dim3(256), (size_t)kWaves * (kPartStride + 768) * sizeof(double),
                               ctx->stream, d_X, ld, d_y, n_feat, n_rows, partials);
        else
            hipLaunchKernelGGL((moments_rowmajor_kernel<T>), dim3(nblocks), dim3(256), (size_t)kWaves * kPartStride * sizeof(double),
                               ctx->stream, d_X, ld, d_y, n_feat, n_rows, partials);
    } else {
        hipLaunchKernelGGL((moments_rowmajor_kernel<T>), dim3(nblocks), dim3(256), (size_t)kWaves * kPartStride * sizeof(double), ctx->stream,
                           d_X, ld, d_y, n_feat, n_rows, partials);
    }
    if (d_moments_f64)
        hipLaunchKernelGGL((moments_finalize_kernel<double>), dim3(kPartSW + 1), dim3(64), 0, ctx->stream, partials, nblocks, n_feat,
                           (double)n_rows, 0, 0, d_moments_f64, (double*)nullptr);
    else
        hipLaunchKernelGGL((moments_finalize_kernel<T>), dim3(kPartSW + 1), dim3(64), 0, ctx->stream, partials, nblocks, n_feat,
                           (double)n_rows, 0, 0, d_moments, (double*)nullptr);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
template int launch_moments_rowmajor<double>(pds_ctx*, const double*, int64_t, const double*, int, int64_t, double*, double*);
template int launch_moments_rowmajor<float>(pds_ctx*, const float*, int64_t, const float*, int, int64_t, float*, double*);

// sum of the per-chunk f64 moment records in chunk order (fixed order: reproducible), cast once
template <typename T>
__global__ void sum_moment_slots_kernel(const double* __restrict__ slots, int nslots, int len, T* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    double s = 0.0;
    for (int k = 0; k < nslots; ++k) s += slots[(size_t)k * len + i];
    out[i] = (T)s;
}
template <typename T>
int launch_sum_moment_slots(pds_ctx* ctx, const double* d_slots, int nslots, int len, T* d_out) {
    hipLaunchKernelGGL((sum_moment_slots_kernel<T>), dim3((len + 255) / 256), dim3(256), 0, ctx->stream, d_slots, nslots, len, d_out);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
// More than 16 features: the IRLS step of the WM = 3 pass as a kernel of its own -- lane = row, run-time loop over the
// columns (coefficients from the scalar cache), writes the row's weight and working response; the weighted wide Gram
// build (moments_wide.hip) then reads them as its weight column and target.  Same arithmetic order as the fused pass.
template <typename T>
__global__ __launch_bounds__(256) void irls_working_wide_kernel(const T* const* __restrict__ cols, int p, int bias, int64_t n,
                                                                const T* __restrict__ beta, IrlsArgs ia, T* __restrict__ w_out,
                                                                T* __restrict__ z_out) {
    const gptr<T> cy = as_global(cols[p]);
    const T b0 = (bias && !ia.init) ? beta[p] : T(0);
    const T ymean = (T)ia.y_mean;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const T yv = cy[r];
        T eta, mu;
        if (ia.init) {
            mu = (ia.variance == 2) ? (yv + T(0.5)) * T(0.5) : (yv + ymean) * T(0.5);
            eta = glm_link<T>(ia.link, mu);
        } else {
            T acc = b0;
            for (int c = 0; c < p; ++c) acc += as_global(cols[c])[r] * beta[c];
            eta = acc;
            mu = glm_inv<T>(ia.link, eta);
        }
        const T d = glm_deriv<T>(ia.link, mu);
        w_out[r] = T(1) / (d * d * glm_var<T>(ia.variance, mu));
        z_out[r] = eta + d * (yv - mu);
    }
}

template <typename T>
int launch_irls_working_wide(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, int bias, const T* d_beta,
                             const IrlsArgs& ia, T* d_w, T* d_z) {
    const int nb = (int)std::min<int64_t>(std::max<int64_t>((n_rows + 255) / 256, 1), (int64_t)ctx->num_cus * 8);
    KernelTimer timer(ctx, kKindPass2);
    hipLaunchKernelGGL((irls_working_wide_kernel<T>), dim3(nb), dim3(256), 0, ctx->stream, dc.d_ptrs, n_feat, bias, n_rows, d_beta, ia,
                       d_w, d_z);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
template int launch_irls_working_wide<double>(pds_ctx*, const DeviceCols<double>&, int, int64_t, int, const double*, const IrlsArgs&, double*, double*);
template int launch_irls_working_wide<float>(pds_ctx*, const DeviceCols<float>&, int, int64_t, int, const float*, const IrlsArgs&, float*, float*);

template int launch_sum_moment_slots<double>(pds_ctx*, const double*, int, int, double*);
template int launch_sum_moment_slots<float>(pds_ctx*, const double*, int, int, float*);

template int launch_moments<double>(pds_ctx*, const DeviceCols<double>&, int, int64_t, bool, double*, const double*, int, double*, double*,
                                    const IrlsArgs*, double*);
template int launch_moments<float>(pds_ctx*, const DeviceCols<float>&, int, int64_t, bool, float*, const float*, int, double*, double*,
                                   const IrlsArgs*, double*);
template int launch_grouped_moments<double>(pds_ctx*, const DeviceCols<double>&, int, const int64_t*, int64_t,
                                            double*, const int32_t*);
template int launch_grouped_moments<float>(pds_ctx*, const DeviceCols<float>&, int, const int64_t*, int64_t,
                                           float*, const int32_t*);

}  // namespace pds
