// moments_wide.hip -- the Gram build for p > 16 features (config 5: elastic net on 1e7 rows x 512 f32
// features, `faer_coordinate_descent`'s one-off X'X / X'y, /root/reference/src/linear/lr/lr_solvers.rs:447-484).
//
// Z = [x_0 .. x_{p-1} | 1 | y] has q = p + 2 columns; A = Z'Z is a tall-skinny SYRK: at p = 512 f32 it is
// 256 flop/B, i.e. MFMA bound (5.3e12 flop vs 20.5 GB), not HBM bound like the p <= 16 kernel.
//   * output tiled 128 x 128 per workgroup (4 waves, each 64 x 64 = 2x2 tiles of v_mfma_f32_32x32x2_f32 or
//     4x4 tiles of v_mfma_f64_16x16x4_f64); only the upper block triangle is computed;
//   * split-K over the row axis (grid.y) so that a few hundred workgroups are in flight; every workgroup
//     writes its f64 partial tile and a fixed-order kernel reduces them (deterministic, no atomics);
//   * the K-panels (128 columns x 128 B of rows) are read with 16-byte loads -- each 8-lane group covers one
//     contiguous 128-byte line of a column -- prefetched into registers one stage ahead and parked in LDS
//     column-major with an odd leading dimension, from where the MFMA operands are conflict-free b32/b64 reads;
//   * f32 splits are at most 8192 rows, accumulated in the f32 matrix-core tile, summed across splits in f64 (the reference
//     accumulates everything in f32).
#include "common.hpp"

#include <type_traits>

namespace pds {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4w __attribute__((ext_vector_type(4), aligned(4)));
typedef double d4w __attribute__((ext_vector_type(4)));
typedef double d2w __attribute__((ext_vector_type(2), aligned(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f2w __attribute__((ext_vector_type(2)));
typedef unsigned u4w __attribute__((ext_vector_type(4)));
typedef unsigned u2w __attribute__((ext_vector_type(2)));

constexpr int kWB = 128;     // block tile edge (columns of Z)
constexpr int kWThreads = 256;

template <typename T>
struct Wide;
template <>
struct Wide<float> {
#ifndef PDS_WIDE_KC
#define PDS_WIDE_KC 32
#endif
    static constexpr int KC = PDS_WIDE_KC;   // rows per stage (128 B per column at 32; 64 -- half the barriers per row, two
                                             // workgroups per CU instead of three -- measured 31.0 ms against 29.5 at config 5)
    static constexpr int CS = KC + 1;        // LDS leading dimension (floats)
    static constexpr int MT = 32;   // MFMA tile edge
    static constexpr int NT = 2;    // tiles per wave edge (64 / MT)
    static constexpr int KS = KC / 2;   // MFMA k-steps per stage
    using vec = f4w;
    static constexpr int VL = 4;
};
// f32 on the bf16 matrix-core path (SPLIT): x = h + m + l exactly, three bf16 planes (round to nearest, so the remainders
// carry random signs); LDS holds each plane k-contiguous per column, 16 pad bytes per column keep the 16-byte operand reads
// (8 lanes = 8 columns, 20 dwords apart) on distinct banks
#ifndef PDS_WIDE_SPLIT_KC
#define PDS_WIDE_SPLIT_KC 32
#endif
constexpr int kSplitKC = PDS_WIDE_SPLIT_KC;          // rows per stage of the SPLIT arithmetic
constexpr int kSplitCSD = kSplitKC / 2 + 4;          // dwords per column and plane
constexpr int kSplitPlane = 128 * kSplitCSD;         // dwords per plane of one panel
// one packed pair of bf16 (round to nearest even) out of two floats; the floats keep the remainders
__device__ __forceinline__ unsigned split_pair(float& a, float& b) {
    const f2w v = {a, b};
    const unsigned bits = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    a -= __builtin_bit_cast(float, bits << 16);
    b -= __builtin_bit_cast(float, bits & 0xffff0000u);
    return bits;
}
// c += a b with a = ah + am + al, b = bh + bm + bl: the six products down to 2^-24 of |a||b| (the three left out are below
// the rounding of the f32 accumulator), smallest first
__device__ __forceinline__ f16v mfma_split(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f16v c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    return c;
}

template <>
struct Wide<double> {
    static constexpr int KC = 16;
    static constexpr int CS = 17;
    static constexpr int MT = 16;
    static constexpr int NT = 4;
    static constexpr int KS = 4;
    using vec = d2w;
    static constexpr int VL = 2;
};

// pair index -> (I, J), I <= J, row-major over the upper triangle of an nb x nb block grid
__host__ __device__ inline void pair_to_ij(int pair, int nb, int& I, int& J) {
    int i = 0;
    while (pair >= nb - i) {
        pair -= nb - i;
        ++i;
    }
    I = i;
    J = i + pair;
}

__host__ __device__ inline int ij_to_pair(int I, int J, int nb) { return I * nb - I * (I - 1) / 2 + (J - I); }

// A last block column holding at most one MFMA tile of valid columns (the [1 | y] tail when p is a multiple of 128)
// must not cost full 128 x 128 tiles of matrix-core time on padding:
// MODE 0: blockIdx.x enumerates the upper triangle of the leading nb_main x nb_main block grid, nothing special.
// MODE 1: "narrow" launch for that tail column, blockIdx.x = I, J = nb - 1: each wave takes 32 rows of the I panel
//         against the first MFMA tile of the J panel (used for f64).
// MODE 2: (f32) like MODE 0, and the tail is fused into the diagonal blocks: there the wave that would compute the
//         redundant lower-left quadrant instead multiplies all 128 rows of the I panel with the tail tile, which is
//         parked in the otherwise unused J half of LDS.  No extra pass over X; tail' tail (a few columns) comes from
//         a one-block-wide MODE 1 launch.
// WEIGHTED: A = Z' diag(w) Z (faer_weighted_lr, lr_solvers.rs:386-409) as the plain Gram of sqrt(w) * Z: every value is
// scaled by sqrt(w_row) on its way into LDS (`sw`, precomputed once per call), which keeps the product symmetric, so
// the diagonal-block and tail shortcuts apply unchanged.
// SPLIT (f32 only): the products run on the bf16 matrix cores as three-plane splits (see mfma_split) -- 6 instructions of
// 32 cycles per 16 rows instead of 8 of 64 on v_mfma_f32_32x32x2_f32, f32 accumulation either way.
template <typename T, int MODE, bool WEIGHTED, bool SPLIT = false>
__global__ __launch_bounds__(kWThreads, (SPLIT && kSplitKC == 16) ? 3 : 2) void moments_wide_kernel(const T* const* __restrict__ cols, int p, int64_t n,
                                                                 int nb, int nb_main, int i_first, int64_t rows_per_split,
                                                                 const T* __restrict__ sw, T* __restrict__ partials) {
    using W = Wide<T>;
    // (MODE 3 / 4 / 5, the compact f64 forms for one block column of up to 128 / 64 / 32 columns: the LDS panel holds
    // 4096 values whatever its shape, so the narrower the frame the more rows a stage takes -- 32 / 64 / 128.  A stage has to be
    // worth a trip to memory: with 16 rows, or 32 rows of a 22-column frame (5.6 KB), the kernel ran at 0.85 TB/s)
    constexpr int KC = SPLIT ? kSplitKC : (MODE == 3 ? 32 : MODE == 4 ? 64 : MODE == 5 ? 128 : W::KC);
    constexpr int CS = MODE >= 3 ? KC + 1 : W::CS, MT = W::MT, NT = W::NT, VL = W::VL;
    constexpr int NCOLS = MODE >= 3 ? 4096 / KC : kWB;  // columns of the LDS panel
    constexpr int PPC = KC / VL;                  // 16-byte pieces per column and stage
    constexpr int NCH = NCOLS * PPC / kWThreads;  // chunks per thread and panel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!SPLIT || (sizeof(T) == 4 && KC % 16 == 0), "the split path is the f32 Gram on bf16 matrix cores");
    T* LI = reinterpret_cast<T*>(smem);
    T* LJ = LI + kWB * CS;
    unsigned* const SI = reinterpret_cast<unsigned*>(smem);  // SPLIT: planes h, m, l of the I panel, then of the J panel
    unsigned* const SJ = SI + 3 * kSplitPlane;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    constexpr bool NARROW = MODE == 1, FUSE = MODE == 2;
    constexpr bool compact = MODE >= 3;  // f64, one block column: see mma_compact below (its own instantiation: sharing one
                                         // with the 4 x 4 form costs 182 spilled registers)
    static_assert(!compact || sizeof(T) == 8 || (SPLIT && MODE == 3), "compact forms: the f64 tile grid, and the f32 split arithmetic (MODE 3)");
    static_assert(!FUSE || (sizeof(T) == 4 && MT * 8 == kWThreads), "tail fusion is laid out for the f32 tile");
    // workgroups are handed to the 8 XCDs round-robin in launch order.  Remap so that all pairs of one row split run
    // on the same XCD back to back: they walk the same rows at the same pace, so each K-panel is pulled from HBM once
    // per split into that XCD's L2 instead of once per pair (every panel is used by ~nb pairs).
    int bx, by;
    {
        const int nx = gridDim.x, ns = gridDim.y;
        const int L = blockIdx.x + nx * blockIdx.y;
        const int ns_full = ns & ~7;
        if (L < nx * ns_full) {
            const int xcd = L & 7, slot = L >> 3;
            bx = slot % nx;
            by = (slot / nx) * 8 + xcd;
        } else {
            const int R = L - nx * ns_full;
            bx = R % nx;
            by = ns_full + R / nx;
        }
    }
    int I, J;
    if constexpr (NARROW) {
        I = i_first + bx;
        J = nb - 1;
    } else {
        pair_to_ij(bx, nb_main, I, J);
    }
    const int pair = ij_to_pair(I, J, nb), npairs = nb * (nb + 1) / 2;
    const bool diag = I == J;
    const int q = p + 2;
    const int64_t r_begin = (int64_t)by * rows_per_split;
    const int64_t r_end = (r_begin + rows_per_split < n) ? r_begin + rows_per_split : n;

    // this thread's 4 chunks per panel: chunk id = tid + 256 u  ->  column id/8, 16-byte piece id%8
    gptr<T> ptrI[NCH];
    gptr<T> ptrJ[NCH];
    unsigned kinds = 0;  // 2 bits per chunk, I panel then J panel: 0 data column, 1 ones column, 2 zero padding
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        const int id = tid + kWThreads * u;
        const int cI = I * kWB + (id / PPC), cJ = ((FUSE && diag) ? nb - 1 : J) * kWB + (id / PPC);
        kinds |= (unsigned)((cI < p || cI == p + 1) ? 0 : (cI == p ? 1 : 2)) << (2 * u);
        kinds |= (unsigned)((cJ < p || cJ == p + 1) ? 0 : (cJ == p ? 1 : 2)) << (2 * (NCH + u));
        ptrI[u] = as_global(cols[cI < p ? cI : p]);  // index p is y in the device table
        ptrJ[u] = as_global(cols[cJ < p ? cJ : p]);
    }
    // block-uniform: every column of the panel is a data column (no ones column, no y, no padding)
    const bool plainI = (I + 1) * kWB <= p, plainJ = (J + 1) * kWB <= p;

    typename W::vec rI[1][NCH], rJ[1][NCH];
    auto load_stage = [&](auto set_c, int64_t row0) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value;
        typename W::vec swv;  // sqrt(w) of this thread's VL rows (the same rows for all of its chunks)
        if constexpr (WEIGHTED) {
            const int64_t r = row0 + (tid % PPC) * VL;
            if (r + VL <= r_end) {
                swv = *reinterpret_cast<gptr<typename W::vec>>(as_global(sw) + r);
            } else {
#pragma unroll
                for (int e = 0; e < VL; ++e) swv[e] = (r + e < r_end) ? sw[r + e] : T(0);
            }
        }
        // the common stage -- all KC rows inside the split, every column of the panel a data column -- is straight 16-byte
        // loads behind wave-uniform tests; the per-lane kind / range selects of the general path cost a wave ~2 000 cycles
        // of scalar branches per stage, which the short matrix-core phase of the SPLIT arithmetic no longer covers
        const bool full = row0 + KC <= r_end;
#pragma unroll
        for (int pnl = 0; pnl < 2; ++pnl) {
            if (pnl == 1 && diag && !FUSE) continue;  // diagonal: J half unused
            if (full && (pnl ? (plainJ && !diag) : plainI)) {
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    const int64_t r = row0 + ((tid + kWThreads * u) % PPC) * VL;
                    typename W::vec v = *reinterpret_cast<gptr<typename W::vec>>((pnl ? ptrJ[u] : ptrI[u]) + r);
                    if constexpr (WEIGHTED) {
#pragma unroll
                        for (int e = 0; e < VL; ++e) v[e] *= swv[e];
                    }
                    if (pnl) rJ[SET][u] = v;
                    else rI[SET][u] = v;
                }
                continue;
            }
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                const int id = tid + kWThreads * u;
                const int64_t r = row0 + (id % PPC) * VL;
                if (pnl == 1 && diag && !(FUSE && id / PPC < 32)) continue;  // diagonal: only the tail tile
                const int kind = (kinds >> (2 * (pnl * NCH + u))) & 3;
                const gptr<T> ptr = pnl ? ptrJ[u] : ptrI[u];
                typename W::vec v;
                if (full) {
                    // a whole stage inside the split (every stage but a ragged last one): a column is a vector load, the ones
                    // column or padding -- one select per chunk instead of the per-element tests below
#pragma unroll
                    for (int e = 0; e < VL; ++e) v[e] = kind == 1 ? T(1) : T(0);
                    if (kind == 0) v = *reinterpret_cast<gptr<typename W::vec>>(ptr + r);
                } else if (kind == 0 && r + VL <= r_end) {
                    v = *reinterpret_cast<gptr<typename W::vec>>(ptr + r);
                } else {
#pragma unroll
                    for (int e = 0; e < VL; ++e) {
                        const bool in = r + e < r_end;
                        v[e] = !in ? T(0) : (kind == 0 ? ptr[r + e] : (kind == 1 ? T(1) : T(0)));
                    }
                }
                if constexpr (WEIGHTED) {
#pragma unroll
                    for (int e = 0; e < VL; ++e) v[e] *= swv[e];
                }
                if (pnl) rJ[SET][u] = v;
                else rI[SET][u] = v;
            }
        }
    };
    auto store_stage = [&](auto set_c) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int id = tid + kWThreads * u;
            const int c = id / PPC, k0 = (id % PPC) * VL;
            if constexpr (SPLIT) {
#pragma unroll
                for (int pnl = 0; pnl < 2; ++pnl) {
                    if (pnl == 1 && diag && !(FUSE && c < 32)) continue;
                    if constexpr (compact) {
                        if (((kinds >> (2 * u)) & 3) == 2) continue;  // padding columns: planes zeroed once, in front of the loop
                    }
                    float v0 = pnl ? rJ[SET][u][0] : rI[SET][u][0], v1 = pnl ? rJ[SET][u][1] : rI[SET][u][1];
                    float v2 = pnl ? rJ[SET][u][2] : rI[SET][u][2], v3 = pnl ? rJ[SET][u][3] : rI[SET][u][3];
                    unsigned* dst = (pnl ? SJ : SI) + c * kSplitCSD + k0 / 2;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        u2w w;
                        w[0] = split_pair(v0, v1);
                        w[1] = split_pair(v2, v3);
                        *reinterpret_cast<u2w*>(dst + pl * kSplitPlane) = w;
                    }
                }
            } else {
                if constexpr (compact) {
                    if (((kinds >> (2 * u)) & 3) == 2) continue;  // padding columns were zeroed once, in front of the loop
                }
#pragma unroll
                for (int e = 0; e < VL; ++e) {
                    LI[c * CS + k0 + e] = rI[SET][u][e];
                    if (!diag || (FUSE && c < 32)) LJ[c * CS + k0 + e] = rJ[SET][u][e];
                }
            }
        }
    };

    // accumulators live in the matrix core's own precision.  f32: a split covers only a few thousand rows
    // (rows_per_split), so the f32 tile error stays ~sqrt(rows) eps and the cross-split sum is done in f64 by the
    // reduce kernel -- keeping 64 f64 accumulators next to the f32 tile cost half the occupancy.
    double accd[sizeof(T) == 4 ? 1 : NT][sizeof(T) == 4 ? 1 : NT][4];
    if constexpr (sizeof(T) == 8) {
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                for (int r = 0; r < 4; ++r) accd[m][nn][r] = 0.0;
    }
    f16v accf[2][2];
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r) accf[m][nn][r] = 0.f;
    }
    const T* PJ = diag ? LI : LJ;
    (void)PJ;
    const bool tailwave = FUSE && diag && wave == 2;
    auto mma_tail = [&]() __attribute__((always_inline)) {
        if constexpr (FUSE && SPLIT) {
            const int li = lane & 31, kq = lane >> 5;
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                bf16x8 b[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    b[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4w*>(SJ + pl * kSplitPlane + li * kSplitCSD + ks * 8 + kq * 4));
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    bf16x8 a[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        a[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4w*>(SI + pl * kSplitPlane + (t * 32 + li) * kSplitCSD + ks * 8 + kq * 4));
                    accf[t >> 1][t & 1] = mfma_split(a, b, accf[t >> 1][t & 1]);
                }
            }
        } else if constexpr (FUSE) {
            const int li = lane & 31, kq = lane >> 5;
#pragma unroll
            for (int ks = 0; ks < W::KS; ++ks) {
                const int k = 2 * ks + kq;
                float a[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) a[t] = LI[(t * 32 + li) * CS + k];
                const float b = LJ[li * CS + k];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    accf[t >> 1][t & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b, accf[t >> 1][t & 1], 0, 0, 0);
            }
        }
    };
    constexpr bool narrow = NARROW;
    const int rb = narrow ? wave * 32 : wr * 64, cb = narrow ? 0 : wc * 64;
    constexpr int NMN = 32 / MT;  // m-tiles per wave in narrow mode
    auto mma_stage = [&](auto nm_c, auto nn_c) __attribute__((always_inline)) {
        constexpr int NM = decltype(nm_c)::value, NN = decltype(nn_c)::value;
        if constexpr (SPLIT) {
            const int li = lane & 31, kq = lane >> 5;
            const unsigned* const SB = diag ? SI : SJ;
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                bf16x8 a[NM][3];
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        a[m][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4w*>(SI + pl * kSplitPlane + (rb + m * 32 + li) * kSplitCSD + ks * 8 + kq * 4));
#pragma unroll
                for (int nn = 0; nn < NN; ++nn) {  // one B tile at a time: 12 operand registers instead of 24
                    bf16x8 b[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        b[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4w*>(SB + pl * kSplitPlane + (cb + nn * 32 + li) * kSplitCSD + ks * 8 + kq * 4));
#pragma unroll
                    for (int m = 0; m < NM; ++m) accf[m][nn] = mfma_split(a[m], b, accf[m][nn]);
                }
            }
        } else if constexpr (sizeof(T) == 4) {
            const int li = lane & 31, kq = lane >> 5;
#pragma unroll
            for (int ks = 0; ks < W::KS; ++ks) {
                const int k = 2 * ks + kq;
                float a[NM], b[NN];
#pragma unroll
                for (int m = 0; m < NM; ++m) a[m] = LI[(rb + m * 32 + li) * CS + k];
#pragma unroll
                for (int nn = 0; nn < NN; ++nn) b[nn] = PJ[(cb + nn * 32 + li) * CS + k];
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int nn = 0; nn < NN; ++nn)
                        accf[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[nn], accf[m][nn], 0, 0, 0);
            }
        } else {
            const int li = lane & 15, kq = lane >> 4;
#pragma unroll
            for (int ks = 0; ks < W::KS; ++ks) {
                const int k = 4 * ks + kq;
                double a[NM], b[NN];
#pragma unroll
                for (int m = 0; m < NM; ++m) a[m] = LI[(rb + m * 16 + li) * CS + k];
#pragma unroll
                for (int nn = 0; nn < NN; ++nn) b[nn] = PJ[(cb + nn * 16 + li) * CS + k];
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int nn = 0; nn < NN; ++nn) {
                        d4w c = {accd[m][nn][0], accd[m][nn][1], accd[m][nn][2], accd[m][nn][3]};
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[nn], c, 0, 0, 0);
                        accd[m][nn][0] = c[0];
                        accd[m][nn][1] = c[1];
                        accd[m][nn][2] = c[2];
                        accd[m][nn][3] = c[3];
                    }
            }
        }
    };
    // f64, ONE block column (17 .. 126 features, the usual width of a hand-built regression): the 128 x 128 tile is mostly
    // padding and all of its useful part sits in the quadrant of wave 0 -- 4.7e11 of 6.5e11 executed flops wasted at 64
    // features and three idle waves.  Compact form (MODE 3): the T = ntile (ntile + 1) / 2 tiles of the upper triangle of the
    // ceil(q / 16)^2 grid of 16 x 16 MFMA tiles are dealt round-robin to the four waves: tile t = wave + 4 s sits in slot s,
    // accumulators accd[s >> 2][s & 3] (at most 9 slots).
    constexpr int TE = sizeof(T) == 8 ? 16 : 32;  // MFMA tile edge (f32: the 32 x 32 tiles of the split arithmetic, 4 rows at most)
    const int ntile = (q + TE - 1) / TE;
    constexpr int kSlots = sizeof(T) == 4 ? 3 : (MODE == 5 ? 1 : (MODE == 4 ? 3 : 9));  // ceil(T / 4) for at most 2 / 4 / 8 tile rows
    int slot_a[kSlots], slot_b[kSlots];  // LDS element offsets of the tile's row / column operand for this lane; -1: empty slot
    if constexpr (compact) {
        const int w0 = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int sidx = 0; sidx < kSlots; ++sidx) {
            int t = w0 + 4 * sidx, r = 0;
            while (r < ntile && t >= ntile - r) {  // row r of the triangle holds ntile - r tiles
                t -= ntile - r;
                ++r;
            }
            const bool ok = r < ntile;
            constexpr int LD = SPLIT ? kSplitCSD : CS;  // elements (f32: dwords of a plane) between two columns of the panel
            slot_a[sidx] = ok ? (r * TE + (lane & (TE - 1))) * LD : -1;
            slot_b[sidx] = ok ? ((r + t) * TE + (lane & (TE - 1))) * LD : -1;
        }
    }
    auto mma_compact = [&]() __attribute__((always_inline)) {
        if constexpr (compact && SPLIT) {
            const int kq = lane >> 5;
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
#pragma unroll
                for (int sidx = 0; sidx < kSlots; ++sidx) {
                    if (__builtin_amdgcn_readfirstlane(slot_a[sidx]) < 0) continue;
                    bf16x8 a[3], b[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        a[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4w*>(SI + pl * kSplitPlane + slot_a[sidx] + ks * 8 + kq * 4));
                        b[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4w*>(SI + pl * kSplitPlane + slot_b[sidx] + ks * 8 + kq * 4));
                    }
                    accf[sidx >> 1][sidx & 1] = mfma_split(a, b, accf[sidx >> 1][sidx & 1]);
                }
            }
        } else if constexpr (compact) {
            const int kq = lane >> 4;
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                const int k = 4 * ks + kq;
#pragma unroll
                for (int sidx = 0; sidx < kSlots; ++sidx) {
                    if (__builtin_amdgcn_readfirstlane(slot_a[sidx]) < 0) continue;  // (wave-uniform: slots fill from the front)
                    const double a = LI[slot_a[sidx] + k], b = LI[slot_b[sidx] + k];
                    d4w c = {accd[sidx >> 2][sidx & 3][0], accd[sidx >> 2][sidx & 3][1], accd[sidx >> 2][sidx & 3][2],
                             accd[sidx >> 2][sidx & 3][3]};
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
                    accd[sidx >> 2][sidx & 3][0] = c[0];
                    accd[sidx >> 2][sidx & 3][1] = c[1];
                    accd[sidx >> 2][sidx & 3][2] = c[2];
                    accd[sidx >> 2][sidx & 3][3] = c[3];
                }
            }
        }
    };
    auto mma_any = [&]() __attribute__((always_inline)) {
        if constexpr (narrow) mma_stage(std::integral_constant<int, NMN>{}, std::integral_constant<int, 1>{});
        else if (tailwave) mma_tail();
        else if constexpr (compact) mma_compact();
        else mma_stage(std::integral_constant<int, NT>{}, std::integral_constant<int, NT>{});
    };
    using set0 = std::integral_constant<int, 0>;
    // ---- the partial tile: P[split][pair][i * 128 + j] (narrow: only the first MFMA tile column)
    T* P = partials + ((int64_t)by * npairs + pair) * (kWB * kWB);
    if constexpr (FUSE) {
        if (tailwave) P = partials + ((int64_t)by * npairs + ij_to_pair(I, nb - 1, nb)) * (kWB * kWB);  // pair (I, nb-1), first tile column
    }
    if constexpr (compact && SPLIT) {  // the padding columns of the panel: zero once (all three planes)
        for (int i = tid; i < 3 * kSplitPlane; i += kWThreads) SI[i] = 0u;
    } else if constexpr (compact) {
        for (int i = tid; i < NCOLS * CS; i += kWThreads) LI[i] = T(0);
    }
    if (r_begin < r_end) load_stage(set0{}, r_begin);
    for (int64_t row0 = r_begin; row0 < r_end; row0 += KC) {
        __syncthreads();  // previous stage's reads are done
        store_stage(set0{});
        __syncthreads();
        if (row0 + KC < r_end) load_stage(set0{}, row0 + KC);
        mma_any();
    }
    if constexpr (FUSE) {
        if (tailwave) {  // rows t * 32.. of pair (I, nb-1), first tile column
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
#pragma unroll
                for (int t = 0; t < 4; ++t) P[(t * 32 + row) * kWB + col] = accf[t >> 1][t & 1][r];
            }
            return;
        }
    }
    if constexpr (compact && SPLIT) {
#pragma unroll
        for (int sidx = 0; sidx < kSlots; ++sidx) {
            if (slot_a[sidx] < 0) continue;
            const int tr = slot_a[sidx] / (32 * kSplitCSD), tc = slot_b[sidx] / (32 * kSplitCSD);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
                P[(tr * 32 + row) * kWB + tc * 32 + col] = accf[sidx >> 1][sidx & 1][r];
            }
        }
        return;
    } else if constexpr (compact) {
#pragma unroll
        for (int sidx = 0; sidx < kSlots; ++sidx) {
            if (slot_a[sidx] < 0) continue;
            // offsets back to tile coordinates: slot_a = (16 r + li) CS, slot_b = (16 c + li) CS
            const int tr = slot_a[sidx] / (16 * CS), tc = slot_b[sidx] / (16 * CS);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                P[(tr * 16 + (lane >> 4) + 4 * r) * kWB + tc * 16 + (lane & 15)] = accd[sidx >> 2][sidx & 3][r];
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
            if (narrow && (m >= NMN || nn >= 1)) continue;
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
                    P[(rb + m * 32 + row) * kWB + cb + nn * 32 + col] = accf[m][nn][r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (lane >> 4) + 4 * r, col = lane & 15;
                    P[(rb + m * 16 + row) * kWB + cb + nn * 16 + col] = accd[m][nn][r];
                }
            }
        }
}


// ------------------------------------------------------------------------------------------------------------------------
// SPLIT arithmetic, 256 x 256 tile per workgroup.  The 128 x 128 kernel above reads two 128-column panels per tile and stage;
// with the products at the bf16 rate THAT is what bounds it: 84 GB go L2 -> L1 per call at config 5 and the CUs sustain
// ~3.8 TB/s of it together (7 B/clk/CU).  Here one workgroup of eight waves owns a 256 x 256 tile (2 x 2 of
// the 128-tiles; the partial tiles keep their layout): 512 columns per stage for four tiles -- half the bytes per flop.
//   * wave (wr, wc) holds rows 64 wr.. of the I superpanel against columns 128 wc.. of the J superpanel: 2 x 4 MFMA tiles, 128
//     accumulator registers, 96 matrix instructions per 32-row stage and wave;
//   * whole 128-byte lines per column and stage (8 lanes x 16 B; 16-row stages with double-buffered planes were measured too:
//     the half lines are fetched twice from L2 -- 512 columns x 128 B outlive the L1 -- and the gain was gone);
//   * diagonal workgroups compute the 10 of their 16 wave tiles that the upper triangle needs; their two idle waves multiply
//     all 256 rows with the [1 | y] tail tile, parked in the unused J half (as MODE 2 does).
// Used when the block grid is even (p = 512, 768, ...) and the tail fits one MFMA tile; everything else stays with the kernel above.
// (Round 5, measured and parked -- tools/experiments/moments_wide_split256_pipelined.hip.txt: the stage as two 16-row halves, a wave
//  converting-and-storing one half of the NEXT stage while it multiplies the other half of the current one, the two waves of a SIMD in
//  complementary order (convert-then-multiply / multiply-then-convert) so that its vector and matrix pipes work at the same time instead of
//  taking turns between the barriers.  Same results, 21.4 against 19.7 ms -- see DESIGN.md 4.6 for the counters and the cut-out variants.)
constexpr int kS2B = 256;                    // tile edge
constexpr int kS2KC = 32;                    // rows per stage
constexpr int kS2CSD = kS2KC / 2 + 4;        // dwords per column and plane (64 B of bf16 + 16 pad: conflict-free 16-byte reads)
constexpr int kS2Plane = 2 * kS2B * kS2CSD;  // dwords per plane: I superpanel then J superpanel
constexpr int kS2Stage = 3 * kS2Plane;       // dwords of the stage buffer (h, m, l): 120 KB, one workgroup per CU
constexpr int kS2Threads = 512;

template <bool WEIGHTED>
__global__ __launch_bounds__(kS2Threads, 1) void moments_wide_split256_kernel(const float* const* __restrict__ cols, int p, int64_t n,
                                                                           int nb, int nb_main, int64_t rows_per_split,
                                                                           const float* __restrict__ sw, float* __restrict__ partials) {
    constexpr int KC = kS2KC, NCH = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* const S = reinterpret_cast<unsigned*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int bx, by;
    {
        const int nx = gridDim.x, ns = gridDim.y;
        const int L = blockIdx.x + nx * blockIdx.y;
        const int ns_full = ns & ~7;
        if (L < nx * ns_full) {
            const int xcd = L & 7, slot = L >> 3;
            bx = slot % nx;
            by = (slot / nx) * 8 + xcd;
        } else {
            const int R = L - nx * ns_full;
            bx = R % nx;
            by = ns_full + R / nx;
        }
    }
    int IS, JS;
    pair_to_ij(bx, nb_main / 2, IS, JS);
    const bool diag = IS == JS;
    const int npairs = nb * (nb + 1) / 2;
    const int64_t r_begin = (int64_t)by * rows_per_split;
    const int64_t r_end = (r_begin + rows_per_split < n) ? r_begin + rows_per_split : n;

    // this thread's eight 16-byte pieces per stage: chunk id = tid + 512 u -> column id / 8 of the 512 (I: u < 4; J: u >= 4),
    // rows 4 (id % 8)..  On the diagonal the J half holds only the tail tile (32 columns of block column nb - 1)
    gptr<float> ptr[NCH];
    unsigned kinds = 0;     // 2 bits per chunk: 0 data column, 1 ones column, 2 zero padding, 3 not loaded
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        const int c = (tid >> 3) + 64 * (u & 3);  // column inside the superpanel
        int col, kind;
        if (u < 4) col = IS * kS2B + c;
        else if (!diag) col = JS * kS2B + c;
        else col = (c < 32) ? (nb - 1) * kWB + c : -1;
        if (col < 0) kind = 3;
        else kind = (col < p || col == p + 1) ? 0 : (col == p ? 1 : 2);
        kinds |= (unsigned)kind << (2 * u);
        ptr[u] = as_global(cols[(col >= 0 && col < p) ? col : p]);  // index p is y in the device table
    }
    const int piece = tid & 7;
    f4w regs[NCH];
    auto load_stage = [&](int64_t row0) __attribute__((always_inline)) {
        const int64_t r = row0 + piece * 4;
        f4w swv;
        if constexpr (WEIGHTED) {
#pragma unroll
            for (int e = 0; e < 4; ++e) swv[e] = (r + e < r_end) ? sw[r + e] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int kind = (kinds >> (2 * u)) & 3;
            if (kind == 3) continue;
            f4w v;
            if (kind == 0 && r + 4 <= r_end) {
                v = *reinterpret_cast<gptr<f4w>>(ptr[u] + r);  // (non-temporal: 21.6 against 21.4 ms)
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = r + e < r_end;
                    v[e] = !in ? 0.f : (kind == 0 ? ptr[u][r + e] : (kind == 1 ? 1.f : 0.f));
                }
            }
            if constexpr (WEIGHTED) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= swv[e];
            }
            regs[u] = v;
        }
    };
    auto store_stage = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            if (((kinds >> (2 * u)) & 3) == 3) continue;
            const int c = (tid >> 3) + 64 * (u & 3) + (u >= 4 ? kS2B : 0);  // column inside the 512
            unsigned* dst = S + c * kS2CSD + piece * 2;
            float v0 = regs[u][0], v1 = regs[u][1], v2 = regs[u][2], v3 = regs[u][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                u2w w;
                w[0] = split_pair(v0, v1);
                w[1] = split_pair(v2, v3);
                *reinterpret_cast<u2w*>(dst + pl * kS2Plane) = w;
            }
        }
    };

    // which of this wave's two 64-column units the upper triangle needs (all of them off the diagonal)
    const bool need0 = !diag || 2 * wc >= wr, need1 = !diag || 2 * wc + 1 >= wr;
    const bool tailwave = diag && nb > nb_main && wc == 0 && wr >= 2;  // rows 128 (wr - 2).. x tail tile
    f16v acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;
    const int li = lane & 31, kq = lane >> 5;
    const unsigned* const BI = S + (tailwave ? (wr - 2) * 128 : wr * 64) * kS2CSD;  // this wave's rows of the I superpanel
    const unsigned* const BJ = S + (diag && !tailwave ? 0 : kS2B) * kS2CSD;        // J superpanel (diagonal: I again; tail: the J half)

    load_stage(r_begin);
    for (int64_t row0 = r_begin; row0 < r_end; row0 += KC) {
        __syncthreads();  // previous stage's reads are done
        store_stage();
        __syncthreads();
        if (row0 + KC < r_end) load_stage(row0 + KC);
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            const int ko = ks * 8 + kq * 4;
            if (tailwave) {
                bf16x8 b[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4w*>(BJ + pl * kS2Plane + li * kS2CSD + ko));
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    bf16x8 a[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        a[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4w*>(BI + pl * kS2Plane + (t * 32 + li) * kS2CSD + ko));
                    acc[t >> 1][t & 1] = mfma_split(a, b, acc[t >> 1][t & 1]);
                }
            } else if (need0 || need1) {
                bf16x8 a[2][3];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        a[m][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4w*>(BI + pl * kS2Plane + (m * 32 + li) * kS2CSD + ko));
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) {
                    if (!((nn < 2) ? need0 : need1)) continue;
                    bf16x8 b[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        b[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4w*>(BJ + pl * kS2Plane + (wc * 128 + nn * 32 + li) * kS2CSD + ko));
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][nn] = mfma_split(a[m], b, acc[m][nn]);
                }
            }
        }
    }
    // ---- partial tiles, P[split][pair of 128-blocks][i * 128 + j]
    const int row0 = 4 * (lane >> 5), col0 = lane & 31;
    if (tailwave) {  // rows t * 32.. of pair (2 IS + wr - 2, nb - 1), first tile column
        float* P = partials + ((int64_t)by * npairs + ij_to_pair(2 * IS + wr - 2, nb - 1, nb)) * (kWB * kWB);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) P[(t * 32 + row0 + (r & 3) + 8 * (r >> 2)) * kWB + col0] = acc[t >> 1][t & 1][r];
        return;
    }
    const int I128 = 2 * IS + (wr >> 1), J128 = 2 * JS + wc;
    if (I128 > J128) return;  // below the diagonal of the block grid (diagonal workgroups only)
    float* P = partials + ((int64_t)by * npairs + ij_to_pair(I128, J128, nb)) * (kWB * kWB);
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) {
        if (!((nn < 2) ? need0 : need1)) continue;  // the redundant lower-left quadrant of a diagonal 128-tile is never read
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                P[((wr & 1) * 64 + m * 32 + row0 + (r & 3) + 8 * (r >> 2)) * kWB + nn * 32 + col0] = acc[m][nn][r];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sqrt_weights_kernel(const T* __restrict__ w, int64_t n, T* __restrict__ sw) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        sw[i] = (T)sqrt((double)w[i]);
}

// The [1 | y] tail against itself -- n, sum y, sum y^2 (weighted: sum w, sum w y, sum w y^2) -- is three sums over one column: a
// grid-stride reduction in f64 instead of the narrow launch of the tile kernel (1.2 ms of barriers at config 5 for 2 x 2 entries).
constexpr int kTailBlocks = 1024;
template <typename T, bool WEIGHTED>
__global__ __launch_bounds__(256) void tail_sums_kernel(const T* __restrict__ y, const T* __restrict__ sw, int64_t n,
                                                        double* __restrict__ partials) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = (double)y[i];
        const double w = WEIGHTED ? (double)sw[i] * (double)sw[i] : 1.0;
        s0 += w;
        s1 = fma(w, v, s1);
        s2 = fma(w * v, v, s2);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s0 += __shfl_xor(s0, o);
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    __shared__ double red[3][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = s0;
        red[1][wave] = s1;
        red[2][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 3) partials[3 * blockIdx.x + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
// fixed-order sum of the block partials (wave k sums component k: lane-strided, then a butterfly), then the four entries of
// the moment matrix (columns p = ones, p + 1 = y)
template <typename T>
__global__ __launch_bounds__(192) void tail_sums_write_kernel(const double* __restrict__ partials, int nblocks, int p, T* __restrict__ out) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += partials[3 * b + k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane != 0) return;
    const int q = p + 2;
    if (k == 0) out[p + (int64_t)p * q] = (T)s;
    if (k == 1) {
        out[p + (int64_t)(p + 1) * q] = (T)s;
        out[(p + 1) + (int64_t)p * q] = (T)s;
    }
    if (k == 2) out[(p + 1) + (int64_t)(p + 1) * q] = (T)s;
}

// out (q x q column-major, symmetric) = sum over splits of the partial tiles, fixed order
template <typename T>
__global__ __launch_bounds__(256) void moments_wide_reduce_kernel(const T* __restrict__ partials, int nsplit,
                                                                  int npairs, int nb, int p, T* __restrict__ out) {
    const int q = p + 2;
    int I, J;
    pair_to_ij(blockIdx.x, nb, I, J);
    for (int e = threadIdx.x + blockIdx.y * blockDim.x; e < kWB * kWB; e += blockDim.x * gridDim.y) {
        const int i = e / kWB, j = e % kWB;
        const int gi = I * kWB + i, gj = J * kWB + j;
        if (gi >= q || gj >= q) continue;
        if (I == J && gi > gj) continue;
        double s = 0.0;
        const T* src = partials + (int64_t)blockIdx.x * (kWB * kWB) + e;
        const int64_t stride = (int64_t)npairs * (kWB * kWB);
        int sidx = 0;
        for (; sidx + 8 <= nsplit; sidx += 8) {  // 8 independent loads in flight, summed in split order
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(sidx + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; sidx < nsplit; ++sidx) s += (double)src[sidx * stride];
        out[gi + (int64_t)gj * q] = (T)s;
        out[gj + (int64_t)gi * q] = (T)s;
    }
}

// split-K plan: enough workgroups to fill the chip, and (f32) few enough rows per split that an f32 accumulator tile
// is safe; splits are multiples of the stage size
template <typename T>
static void wide_split(int num_cus, int npairs, int64_t n_rows, int& nsplit, int64_t& rows_per_split) {
    using W = Wide<T>;
    int64_t want = std::max<int64_t>(1, ((int64_t)num_cus * 4 + npairs - 1) / npairs);
    if (sizeof(T) == 4) want = std::max<int64_t>(want, (n_rows + 8191) / 8192);
    want = std::min<int64_t>(want, std::max<int64_t>(1, (n_rows + 1023) / 1024));
    rows_per_split = (n_rows + want - 1) / want;
    rows_per_split = ((rows_per_split + W::KC - 1) / W::KC) * W::KC;
    nsplit = (int)((n_rows + rows_per_split - 1) / rows_per_split);
}

template <typename T, bool WEIGHTED, bool SPLIT>
static int launch_moments_wide_w(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, T* d_moments) {
    using W = Wide<T>;
    const int q = n_feat + 2;
    const int nb = (q + kWB - 1) / kWB;
    const int npairs = nb * (nb + 1) / 2;
    int nsplit;
    int64_t rows_per_split;
    wide_split<T>(ctx->num_cus, npairs, n_rows, nsplit, rows_per_split);
    const size_t part_bytes = (size_t)nsplit * npairs * kWB * kWB * sizeof(T);
    T* partials = reinterpret_cast<T*>(ws_take(ctx, part_bytes));
    if (ctx->ws_used > ctx->ws.bytes) return fail(PDS_ERR_INVALID, "internal: workspace for the wide Gram build was not reserved");
    const size_t lds = SPLIT ? (size_t)6 * kSplitPlane * sizeof(unsigned) : (size_t)2 * kWB * W::CS * sizeof(T);
    T* d_sw = nullptr;
    if constexpr (WEIGHTED) d_sw = reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T)));
    if (ctx->ws_used > ctx->ws.bytes) return fail(PDS_ERR_INVALID, "internal: workspace for the wide Gram build was not reserved");
    if (lds > 64 * 1024) {
        PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&moments_wide_kernel<T, 0, WEIGHTED, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&moments_wide_kernel<T, 1, WEIGHTED, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if constexpr (sizeof(T) == 4)
            PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&moments_wide_kernel<T, 2, WEIGHTED, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    KernelTimer timer(ctx, kKindMoments);
    if constexpr (WEIGHTED)
        hipLaunchKernelGGL((sqrt_weights_kernel<T>), dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, dc.h_ptrs[n_feat + 1], n_rows,
                           d_sw);
    const bool tail_narrow = q - (nb - 1) * kWB <= W::MT;
    const int nb_main = tail_narrow ? nb - 1 : nb;
    const dim3 grid_main(nb_main * (nb_main + 1) / 2, nsplit);
    bool fused = false, tail_by_sums = false, done1 = false;
    if constexpr (sizeof(T) == 4 && SPLIT) {
        if (nb == 1) {  // 17 .. 126 features, split arithmetic: the compact 32 x 32 tile grid
            hipLaunchKernelGGL((moments_wide_kernel<T, 3, WEIGHTED, true>), dim3(1, nsplit), dim3(kWThreads), lds, ctx->stream, dc.d_ptrs,
                               n_feat, n_rows, nb, 1, 0, rows_per_split, d_sw, partials);
            done1 = true;
        }
    }
    if constexpr (sizeof(T) == 4) {
        if (tail_narrow && nb_main > 0 && !done1) {
            fused = true;
            const char* e128 = dev_env("PDS_WIDE_TILE128");  // A/B: keep the 128 x 128 tile
            if (SPLIT && nb_main >= 2 && nb_main % 2 == 0 && !(e128 && e128[0] == '1')) {
                constexpr int lds256 = kS2Stage * (int)sizeof(unsigned);
                auto kern256 = &moments_wide_split256_kernel<WEIGHTED>;
                PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern256), hipFuncAttributeMaxDynamicSharedMemorySize, lds256));
                const int nbs = nb_main / 2;
                hipLaunchKernelGGL(kern256, dim3(nbs * (nbs + 1) / 2, nsplit), dim3(kS2Threads), lds256,
                                   ctx->stream, dc.d_ptrs, n_feat, n_rows, nb, nb_main, rows_per_split, d_sw, partials);
            } else {
                hipLaunchKernelGGL((moments_wide_kernel<T, 2, WEIGHTED, SPLIT>), grid_main, dim3(kWThreads), lds, ctx->stream, dc.d_ptrs,
                                   n_feat, n_rows, nb, nb_main, 0, rows_per_split, d_sw, partials);
            }
            tail_by_sums = q - (nb - 1) * kWB == 2;  // the tail is exactly [1 | y] (p a multiple of 128)
            if (!tail_by_sums)
                hipLaunchKernelGGL((moments_wide_kernel<T, 1, WEIGHTED, SPLIT>), dim3(1, nsplit), dim3(kWThreads), lds, ctx->stream,
                                   dc.d_ptrs, n_feat, n_rows, nb, nb_main, nb - 1, rows_per_split, d_sw, partials);
        }
    }
    if (!fused && !done1) {
        bool done = false;
        if constexpr (sizeof(T) == 8) {
            if (nb == 1) {  // 17 .. 126 features: the compact tile grid, the more rows per stage the narrower the frame
                if (q <= 32)
                    hipLaunchKernelGGL((moments_wide_kernel<T, 5, WEIGHTED, false>), grid_main, dim3(kWThreads), lds, ctx->stream, dc.d_ptrs,
                                       n_feat, n_rows, nb, nb_main, 0, rows_per_split, d_sw, partials);
                else if (q <= 64)
                    hipLaunchKernelGGL((moments_wide_kernel<T, 4, WEIGHTED, false>), grid_main, dim3(kWThreads), lds, ctx->stream, dc.d_ptrs,
                                       n_feat, n_rows, nb, nb_main, 0, rows_per_split, d_sw, partials);
                else
                    hipLaunchKernelGGL((moments_wide_kernel<T, 3, WEIGHTED, false>), grid_main, dim3(kWThreads), lds, ctx->stream, dc.d_ptrs,
                                       n_feat, n_rows, nb, nb_main, 0, rows_per_split, d_sw, partials);
                done = true;
            }
        }
        if (nb_main > 0 && !done)
            hipLaunchKernelGGL((moments_wide_kernel<T, 0, WEIGHTED, SPLIT>), grid_main, dim3(kWThreads), lds, ctx->stream, dc.d_ptrs,
                               n_feat, n_rows, nb, nb_main, 0, rows_per_split, d_sw, partials);
        if (tail_narrow)
            hipLaunchKernelGGL((moments_wide_kernel<T, 1, WEIGHTED, SPLIT>), dim3(nb, nsplit), dim3(kWThreads), lds, ctx->stream,
                               dc.d_ptrs, n_feat, n_rows, nb, nb_main, 0, rows_per_split, d_sw, partials);
    }
    hipLaunchKernelGGL((moments_wide_reduce_kernel<T>), dim3(npairs, 64), dim3(256), 0, ctx->stream, partials, nsplit,
                       npairs, nb, n_feat, d_moments);
    if (tail_by_sums) {  // (the reduce kernel has just written the unset partials of pair (nb - 1, nb - 1) there: replaced)
        double* tp = reinterpret_cast<double*>(partials);  // the partial tiles are consumed: reuse their head
        const int nblk = (int)std::min<int64_t>(kTailBlocks, std::max<int64_t>(1, (n_rows + 255) / 256));
        hipLaunchKernelGGL((tail_sums_kernel<T, WEIGHTED>), dim3(nblk), dim3(256), 0, ctx->stream, dc.h_ptrs[n_feat], d_sw, n_rows, tp);
        hipLaunchKernelGGL((tail_sums_write_kernel<T>), dim3(1), dim3(192), 0, ctx->stream, tp, nblk, n_feat, d_moments);
    }
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

template <typename T>
int launch_moments_wide(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, bool weighted, T* d_moments) {
    {
        // 17 .. 64 features: the streaming kernel of moments_mid.hip, both precisions (PDS_MID_GRAM=0 keeps the compact tile forms: A/B)
        const char* e = dev_env("PDS_MID_GRAM");
        if (n_feat <= 64 && !(e && e[0] == '0')) return launch_moments_mid<T>(ctx, dc, n_feat, n_rows, weighted, d_moments);
    }
    if constexpr (sizeof(T) == 4) {
        // f32: products on the bf16 matrix cores as three-plane splits (2.7x the f32 matrix-core rate at f32 accuracy);
        // the context option "wide_f32_native" keeps v_mfma_f32_32x32x2_f32 (A/B, and the exact-fmaf-chain arithmetic)
        if (!ctx->opt_wide_f32_native)  // (context option "wide_f32_native"; default from PDS_WIDE_F32_NATIVE at pds_ctx_create)
            return weighted ? launch_moments_wide_w<T, true, true>(ctx, dc, n_feat, n_rows, d_moments)
                            : launch_moments_wide_w<T, false, true>(ctx, dc, n_feat, n_rows, d_moments);
    }
    return weighted ? launch_moments_wide_w<T, true, false>(ctx, dc, n_feat, n_rows, d_moments)
                    : launch_moments_wide_w<T, false, false>(ctx, dc, n_feat, n_rows, d_moments);
}

size_t moments_wide_workspace(int num_cus, int n_feat, int64_t n_rows, bool weighted) {
    const int q = n_feat + 2;
    const int nb = (q + kWB - 1) / kWB;
    const int npairs = nb * (nb + 1) / 2;
    int ns;
    int64_t rps;
    wide_split<float>(num_cus, npairs, n_rows, ns, rps);  // the f32 plan has the most splits; 8 B covers both types
    const size_t wide = (size_t)(ns + 1) * npairs * kWB * kWB * sizeof(double) + 4096 + (weighted ? (size_t)n_rows * sizeof(double) + 512 : 0);
    return std::max(wide, n_feat <= 64 ? moments_mid_workspace(num_cus) : (size_t)0);
}

template int launch_moments_wide<double>(pds_ctx*, const DeviceCols<double>&, int, int64_t, bool, double*);
template int launch_moments_wide<float>(pds_ctx*, const DeviceCols<float>&, int, int64_t, bool, float*);

}  // namespace pds
