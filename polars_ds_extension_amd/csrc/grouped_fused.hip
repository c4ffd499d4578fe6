// grouped_fused.hip -- `group_by(key).agg(pds.lin_reg(...))` as ONE streaming kernel: the rows are read
// exactly like the single-regression Gram build (moments.hip: 128-row tiles, 16 B per lane straight down each
// column, next tile prefetched in registers, MFMA consumption from a wave-private LDS tile) and the running
// accumulators are simply CUT at group boundaries.  Each finished group's normal equations are transposed
// through a 2.6 KB LDS scratch into the registers of a 16/8/4-lane sub-group; when all sub-groups of the wave
// hold a pending system they are solved side by side in registers (solve_reg_dev.hpp) and the coefficients are
// written.  The per-group moment records of the two-kernel pipeline (2.6 KB written + re-read per group at
// p = 16) never exist, HBM traffic is input + coefficients only, and memory-bound tile streaming of some waves
// overlaps the VALU-bound solves of others on the same CU.
//
// Work split: wave w owns the groups whose first row lies in [w N / W, (w+1) N / W) -- balanced in ROWS
// (binary search in the offsets), so skewed group sizes do not unbalance the stream.
//
// Solver: with the rank gate on (the default of pds.lin_reg) the systems are factored by Cholesky in registers
// (breakdown or rel. determinant <= tol => null, exactly the reference's `choleskey` rule, lr_solvers.rs:369-380);
// accepted systems are well conditioned and the reference's three solvers agree on them to rounding.  With the
// gate off (singular_x_tol = 0) the pivoted Householder QR runs instead, because only pivoting reproduces the
// reference's finite answers on rank-deficient groups.  PDS_GROUPED_PIVOTED=1 forces QR everywhere.
#include <type_traits>

#include "moments_dev.hpp"
#include "solve_reg_dev.hpp"

namespace pds {

// -DPDS_PROFILE_PHASES: per-phase shader-clock sums of every wave (development; tools/phase_profile.py reads them)
#ifdef PDS_PROFILE_PHASES
__device__ unsigned long long g_phase_cycles[8];
#define PDS_T0() const unsigned long long _t0 = __builtin_amdgcn_s_memtime()
#define PDS_T1(k) prof[k] += __builtin_amdgcn_s_memtime() - _t0
#else
#define PDS_T0() do {} while (0)
#define PDS_T1(k) do {} while (0)
#endif

constexpr int kFQ = 18;                  // LDS scratch moment matrix: indices 0..15 features, 16 bias, 17 y
constexpr int kFM = kFQ * kFQ;           // doubles
constexpr int kFTile = 17 * kColStride;  // 16 feature slots + y
constexpr int kFWaveLds = kFTile + kFM * 8;

// consume rows [rel0, rel1) of the tile (relative row indices, 0 <= rel0 < rel1 <= TR) into `a`
// BIAS = false drops the column sums / sum y (only the bias row of the normal equations needs them); y'y is never
// needed by the solve.  These side sums are VALU work per 4-row step, and the fused kernel is VALU-issue bound.
template <typename T, bool BIAS>
__device__ __forceinline__ void consume_range(const char* wl, int lane, int rel0, int rel1, WaveAcc& a) {
    const int f = lane & 15, q = lane >> 4;
    const T* xcol = reinterpret_cast<const T*>(wl + f * kColStride) + q;
    const T* ycol = reinterpret_cast<const T*>(wl + kSlotY * kColStride) + q;
    int s0 = rel0 >> 2;
    const int s1 = (rel1 + 3) >> 2;  // exclusive
    using Acc = typename Tile<T>::acc;
    Acc acc;
    double xy, cs, yy, ys;
    if constexpr (sizeof(T) == 8) {
        acc = Acc{a.d[0], a.d[1], a.d[2], a.d[3]};
        xy = a.xy; cs = a.cs; yy = a.yy; ys = a.ys;
    } else {
        acc = Acc{0, 0, 0, 0};
        xy = cs = yy = ys = 0.0;
    }
    T fxy = 0, fcs = 0, fyy = 0, fys = 0;  // f32 path accumulates the side sums in f32 per call, like the tile kernel
    auto step_v = [&](T x, T yv) __attribute__((always_inline)) {
        acc = Tile<T>::mfma(x, x, acc);
        if constexpr (sizeof(T) == 8) {
            xy = fma(x, yv, xy);
            if constexpr (BIAS) {
                cs += x;
                ys += yv;
            }
        } else {
            fxy = fmaf(x, yv, fxy);
            if constexpr (BIAS) {
                fcs += x;
                fys += yv;
            }
        }
    };
    auto step = [&](int s, bool masked, int lo, int hi) __attribute__((always_inline)) {
        T x = xcol[4 * s];
        T yv = ycol[4 * s];
        if (masked) {
            const bool in = q >= lo && q < hi;
            x = in ? x : T(0);
            yv = in ? yv : T(0);
        }
        acc = Tile<T>::mfma(x, x, acc);
        if constexpr (sizeof(T) == 8) {
            xy = fma(x, yv, xy);
            if constexpr (BIAS) {
                cs += x;
                ys += yv;
            }
        } else {
            fxy = fmaf(x, yv, fxy);
            if constexpr (BIAS) {
                fcs += x;
                fys += yv;
            }
        }
    };
    // head: a first step that starts inside a 4-row group
    if (rel0 & 3) {
        const int lo = rel0 & 3;
        const int hi = min(rel1 - 4 * s0, 4);
        step(s0, true, lo, hi);
        ++s0;
    }
    // body: whole steps
    const int sfull = rel1 >> 2;  // steps [s0, sfull) are complete
    // (software pipelined like consume_tile: operands of step s+2 are in flight while step s multiplies)
    // four steps per iteration: eight independent LDS reads first, then four back-to-back matrix instructions, so
    // the ds_read latency is paid once per 16 rows (left as a 1-step loop the compiler put a full lgkmcnt wait in
    // front of every MFMA)
    int s = s0;
    for (; s + 4 <= sfull; s += 4) {
        const T x0 = xcol[4 * s], x1 = xcol[4 * s + 4], x2 = xcol[4 * s + 8], x3 = xcol[4 * s + 12];
        const T y0 = ycol[4 * s], y1 = ycol[4 * s + 4], y2 = ycol[4 * s + 8], y3 = ycol[4 * s + 12];
        step_v(x0, y0);
        step_v(x1, y1);
        step_v(x2, y2);
        step_v(x3, y3);
    }
    // (fetching the next iteration's operands ahead of these matrix instructions was measured slower with a register
    //  rotation -- sixteen moves per iteration -- and no faster with two ping-pong operand sets: 2.43 ms either way)
    for (; s < sfull; ++s) step_v(xcol[4 * s], ycol[4 * s]);
    // tail: a last step that ends inside a 4-row group (and was not already the head step)
    if ((rel1 & 3) && sfull >= s0 && sfull < s1) step(sfull, true, 0, rel1 & 3);
    if constexpr (sizeof(T) == 8) {
        a.d[0] = acc[0]; a.d[1] = acc[1]; a.d[2] = acc[2]; a.d[3] = acc[3];
        a.xy = xy; a.cs = cs; a.yy = yy; a.ys = ys;
    } else {
        a.d[0] += (double)acc[0]; a.d[1] += (double)acc[1]; a.d[2] += (double)acc[2]; a.d[3] += (double)acc[3];
        a.xy += (double)fxy; a.cs += (double)fcs; a.yy += (double)fyy; a.ys += (double)fys;
    }
}

// p <= 8: TWO 4-row slabs per matrix instruction.  Operand columns 0-7 carry the features of rows 8s+q, columns 8-15
// the same features of rows 8s+4+q; the product's two diagonal 8 x 8 blocks are the Gram contributions of the two
// slabs (the off-diagonal blocks are cross terms nobody reads), folded together at flush time.  Half the MFMAs per
// row -- at 8 features the f64 matrix pipe (86 clk per 16x16x4) is as scarce as HBM bandwidth.
template <typename T, bool BIAS>
__device__ __forceinline__ void consume_range_pack(const char* wl, int lane, int rel0, int rel1, WaveAcc& a) {
    const int f = lane & 15, q = lane >> 4;
    const int roff = q + 4 * (f >> 3);  // this lane's row inside an 8-row step
    const T* xcol = reinterpret_cast<const T*>(wl + (f & 7) * kColStride) + roff;
    const T* ycol = reinterpret_cast<const T*>(wl + kSlotY * kColStride) + roff;
    int s0 = rel0 >> 3;
    const int s1 = (rel1 + 7) >> 3;  // exclusive
    using Acc = typename Tile<T>::acc;
    Acc acc;
    double xy, cs, yy, ys;
    if constexpr (sizeof(T) == 8) {
        acc = Acc{a.d[0], a.d[1], a.d[2], a.d[3]};
        xy = a.xy; cs = a.cs; yy = a.yy; ys = a.ys;
    } else {
        acc = Acc{0, 0, 0, 0};
        xy = cs = yy = ys = 0.0;
    }
    T fxy = 0, fcs = 0, fyy = 0, fys = 0;
    auto step_v = [&](T x, T yv) __attribute__((always_inline)) {
        acc = Tile<T>::mfma(x, x, acc);
        if constexpr (sizeof(T) == 8) {
            xy = fma(x, yv, xy);
            if constexpr (BIAS) {
                cs += x;
                ys += yv;
            }
        } else {
            fxy = fmaf(x, yv, fxy);
            if constexpr (BIAS) {
                fcs += x;
                fys += yv;
            }
        }
    };
    auto step_m = [&](int s) __attribute__((always_inline)) {  // a step that straddles rel0 and / or rel1
        const int r = 8 * s + roff;
        const bool in = r >= rel0 && r < rel1;
        const T x = in ? xcol[8 * s] : T(0);
        const T yv = in ? ycol[8 * s] : T(0);
        step_v(x, yv);
    };
    if (rel0 & 7) {
        step_m(s0);
        ++s0;
    }
    const int sfull = rel1 >> 3;  // steps [s0, sfull) are complete
    int s = s0;
    for (; s + 4 <= sfull; s += 4) {
        const T x0 = xcol[8 * s], x1 = xcol[8 * s + 8], x2 = xcol[8 * s + 16], x3 = xcol[8 * s + 24];
        const T y0 = ycol[8 * s], y1 = ycol[8 * s + 8], y2 = ycol[8 * s + 16], y3 = ycol[8 * s + 24];
        step_v(x0, y0);
        step_v(x1, y1);
        step_v(x2, y2);
        step_v(x3, y3);
    }
    for (; s < sfull; ++s) step_v(xcol[8 * s], ycol[8 * s]);
    if ((rel1 & 7) && sfull >= s0 && sfull < s1) step_m(sfull);
    if constexpr (sizeof(T) == 8) {
        a.d[0] = acc[0]; a.d[1] = acc[1]; a.d[2] = acc[2]; a.d[3] = acc[3];
        a.xy = xy; a.cs = cs; a.yy = yy; a.ys = ys;
    } else {
        a.d[0] += (double)acc[0]; a.d[1] += (double)acc[1]; a.d[2] += (double)acc[2]; a.d[3] += (double)acc[3];
        a.xy += (double)fxy; a.cs += (double)fcs; a.yy += (double)fyy; a.ys += (double)fys;
    }
}

__device__ __forceinline__ int64_t lower_bound_i64(const int64_t* __restrict__ a, int64_t n, int64_t key) {
    int64_t lo = 0, hi = n;  // first index with a[idx] >= key
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// FULLP: p' == LPS, i.e. the feature count is known at compile time (the 16- and 8-feature frames of the headline configs,
// 15 / 7 / 3 features + bias): the per-column `c < p` branches of the tile load / store (17 scalar compares + branches per
// tile, each re-reading a spilled SGPR) and the solver's per-step `K < p'` branches fold away -- 2.86 -> 2.63 ms at 16
// features, 1.89 -> 1.52 ms at 8
template <typename T, int LPS, bool CHOL, bool BIAS, int PC>
__global__ __launch_bounds__(64) void grouped_stream_kernel(const T* const* __restrict__ cols, int p_arg,
                                                            const int64_t* __restrict__ offsets, int64_t n_groups,
                                                            int64_t n_rows, SolveRegDev sp, T* __restrict__ coeffs,
                                                            uint8_t* __restrict__ flags, int32_t* __restrict__ mark_list,
                                                            unsigned* __restrict__ mark_count, unsigned* __restrict__ mark_host_flag) {
    constexpr int SPW = 64 / LPS;
    constexpr int RPL = Tile<T>::RPL;
    constexpr int TR = 64 * RPL;
    constexpr int NA = CHOL ? LPS + 1 : LPS;
    constexpr bool PACK = LPS <= 8;        // p <= 8: two row slabs per MFMA (consume_range_pack)
    constexpr int ZSLOT = PACK ? 7 : 15;   // a scratch row / column that holds exact zeros whenever it is read
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    char* wl = smem;
    double* Msc = reinterpret_cast<double*>(smem + kFTile);
    // BIAS: the intercept never takes a solver lane.  The group's normal equations are centred when they leave LDS
    // (G_ij - s_i s_j / n, c_j - s_j sum(y) / n with the column sums s the kernel carries anyway), the p x p system is
    // solved, and b0 = (sum(y) - s . beta) / n.  det([X 1]'[X 1]) = n det(Xc'Xc) and the diagonal products differ by the
    // same n, so the reference's rank gate on the augmented matrix is the gate on the centred pivots over the UNcentred
    // diagonal -- same accept / reject rule.  16 features + bias stay on this kernel, 8 + bias on the packed one.
    constexpr bool FULLP = PC != 0;     // PC: the feature count as a compile-time constant (0 = run-time p_arg)
    static_assert(PC >= 0 && PC <= LPS, "PC is a feature count of this kernel size");
    const int p = FULLP ? PC : p_arg;   // features = solver lanes in use
    const int pp = p;                   // (solver size)
    const int pout = p + (BIAS ? 1 : 0);
    const int sub = lane / LPS, j_in = lane % LPS;

    // ---- this wave's groups: balanced in rows
    const int64_t W = gridDim.x, w = blockIdx.x;
    const int64_t base_row = offsets[0];
    const int64_t total = offsets[n_groups] - base_row;
    const int64_t gl = (w == 0) ? 0 : lower_bound_i64(offsets, n_groups, base_row + (total * w) / W);
    const int64_t gh = (w == W - 1) ? n_groups : lower_bound_i64(offsets, n_groups, base_row + (total * (w + 1)) / W);
    if (gl >= gh) return;
    const int64_t rlo = offsets[gl], rhi = offsets[gh];

    // zero the unused feature slots of the tile once
    {
        typename Tile<T>::vec z;
#pragma unroll
        for (int e = 0; e < RPL; ++e) z[e] = T(0);
        for (int c = p; c < 16; ++c) *reinterpret_cast<typename Tile<T>::vec*>(wl + c * kColStride + lane * 16) = z;
    }
#ifdef PDS_PROFILE_PHASES
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    // pending systems (one per sub-group), held in registers in solver layout
    double a_p[NA];
    double b_p[CHOL ? 1 : LPS];
    double dj_p = 1.0;
    double cs_p = 0.0, ys_p = 0.0, n_p = 1.0;  // BIAS: column sum of the lane's feature, sum(y), rows of the group
    bool null_p = false;
    int npend = 0;
    int64_t gbase = gl;  // group id of pending system 0

    auto solve_pending = [&]() __attribute__((always_inline)) {
        // the solver compares the lane index against 16 constants; hiding it behind an empty asm makes the compiler
        // rebuild those lane masks here instead of keeping 30+ SGPR pairs alive (and spilled) across the whole kernel
        int j = j_in;
        asm volatile("" : "+v"(j));
        const bool colv = j < pp;
        const int64_t sys = gbase + sub;
        const bool live = sub < npend;
        const bool few = null_p;  // fewer rows than coefficients: null whatever the solver says
        bool is_null = null_p;
        bool suspect = false;
        double zj = 0.0;
        int pj = j;
        PDS_T0();
        if constexpr (CHOL) {
            SolveRegDev spk = sp;
            spk.pp = p;  // (a compile-time constant under PC != 0: the per-step `K < p'` branches fold away)
            spk.p = spk.pp;
            spk.gate_on = 1;           // this kernel only exists for the gated Cholesky (launch_stream_lps)
            chol_core<LPS>(a_p, dj_p, j, spk, is_null, zj, &suspect);
            // marginal / gated / broken-down systems are left to the pivoted-QR pass behind this kernel (flag 2)
            suspect = suspect && !few;
            is_null = is_null || suspect;
        } else {
            solve_core<LPS>(a_p, b_p, dj_p, j, lane, sp, is_null, pj, zj);
        }
        if (live && colv) coeffs[sys * (int64_t)pout + pj] = is_null ? (T)__builtin_nan("") : (T)zj;
        if constexpr (BIAS) {
            const double sb = Grp<LPS>::sum(colv ? cs_p * zj : 0.0);
            const double b0 = (ys_p - sb) / n_p;
            if (live && j == 0) coeffs[sys * (int64_t)pout + p] = is_null ? (T)__builtin_nan("") : (T)b0;
        }
        if (live && j == 0 && flags) flags[sys] = suspect ? 2 : (is_null ? 1 : 0);
        if constexpr (CHOL) {
            // systems left to the pivoted-QR pass are appended to a list (one atomic per wave and batch of solves -- and none
            // at all in the common case of no marked system); the first marking wave also raises a word in host-mapped
            // memory, which is all the host looks at after the kernel: no counter read-back, no extra launch unless needed
            const bool mk = suspect && live && j == 0;
            const unsigned long long mm = __ballot(mk);
            if (mm != 0ull && mark_list) {
                const int leader = __ffsll((long long)mm) - 1;
                unsigned slot = 0;
                if (lane == leader) {
                    slot = atomicAdd(mark_count, (unsigned)__popcll(mm));
                    __hip_atomic_store(mark_host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                slot = __shfl(slot, leader);
                if (mk) mark_list[slot + __popcll(mm & ((1ull << lane) - 1ull))] = (int32_t)sys;
            }
        }
        PDS_T1(3);
        gbase += npend;
        npend = 0;
    };

    WaveAcc acc;
    zero_acc(acc);
    int64_t g = gl;
    // group bounds are carried, not re-loaded: gs = start of group g, ge = its end, ge_next = end of group g+1
    // (fetched one group ahead, so the dependent global load sits under a whole group of work instead of in
    // front of every boundary decision)
    // The loop below decides in tile-relative 32-bit numbers (scalar compares): 64-bit `<` has no scalar form and every
    // one of them was a VALU compare + vcc branch on the per-group path.  g_left = groups of this wave not yet flushed.
    int64_t gs = rlo;
    int64_t ge = offsets[g + 1];
    uint32_t g_left = (uint32_t)(gh - gl);  // launch_grouped_fused keeps n_groups below 2^31
    int64_t ge_next = offsets[g_left >= 2 ? g + 2 : gh];
    int64_t pos = rlo;

    auto flush_group = [&]() __attribute__((always_inline)) {
        // ---- normal equations of group g -> LDS scratch (full symmetric square, bias at 16, y at 17)
        PDS_T0();
        const uint64_t ng = (uint64_t)(ge - gs);
        const bool too_few = (uint32_t)(ng >> 32) == 0u && (uint32_t)ng < (uint32_t)pout;
        {
            const int jj = lane & 15;
            // packed kernels, f64: the second slab's diagonal block sits in the same row slot eight lanes to the right
            // (D[i + 8][j + 8]: register r + 2, lane + 8), so it is folded onto the first with a DPP row shift before
            // anything goes to LDS (the fold through LDS was two more synchronisations per group)
            constexpr bool RFOLD = PACK && sizeof(T) == 8;
            auto shl8 = [](double v) __attribute__((always_inline)) {
                int lo = __double2loint(v), hi = __double2hiint(v);
                lo = __builtin_amdgcn_update_dpp(0, lo, 0x108 /*row_shl:8*/, 0xf, 0xf, true);
                hi = __builtin_amdgcn_update_dpp(0, hi, 0x108, 0xf, 0xf, true);
                return __hiloint2double(hi, lo);
            };
            if constexpr (RFOLD) {
                const double d0 = acc.d[0] + shl8(acc.d[2]), d1 = acc.d[1] + shl8(acc.d[3]);
                if (jj < 8) {
                    Msc[Tile<T>::drow(lane, 0) + kFQ * jj] = d0;
                    Msc[Tile<T>::drow(lane, 1) + kFQ * jj] = d1;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) Msc[Tile<T>::drow(lane, r) + kFQ * jj] = acc.d[r];
            }
            double xy = xor_sum_q(acc.xy);
            if constexpr (RFOLD) xy += shl8(xy);
            if (lane < (RFOLD ? 8 : 16)) Msc[lane + kFQ * 17] = xy;
            if constexpr (BIAS) {
                double cs = xor_sum_q(acc.cs), ys = xor_sum_q(acc.ys);
                if constexpr (RFOLD) {
                    cs += shl8(cs);
                    ys += shl8(ys);
                }
                if (lane < (RFOLD ? 8 : 16)) Msc[lane + kFQ * 16] = cs;
                if (lane == 0) {
                    Msc[16 + kFQ * 16] = (double)ng;
                    Msc[16 + kFQ * 17] = ys;
                }
                if (PACK && !RFOLD && lane == 8) Msc[17 + kFQ * 17] = ys;  // sum y over the second row slab
            }
        }
        if constexpr (PACK && sizeof(T) != 8) {  // f32 tile layout: fold the second slab's block (and side sums) through LDS
            PDS_WAVE_LDS_SYNC();
            const int fi = lane & 7, fj = lane >> 3;
            const double dsum = Msc[fi + kFQ * fj] + Msc[fi + 8 + kFQ * (fj + 8)];
            double xys = 0.0, css = 0.0, yss = 0.0;
            if (lane < 8) {
                xys = Msc[lane + kFQ * 17] + Msc[lane + 8 + kFQ * 17];
                if constexpr (BIAS) css = Msc[lane + kFQ * 16] + Msc[lane + 8 + kFQ * 16];
            }
            if (BIAS && lane == 0) yss = Msc[16 + kFQ * 17] + Msc[17 + kFQ * 17];
            PDS_WAVE_LDS_SYNC();
            Msc[fi + kFQ * fj] = dsum;
            if (lane < 8) {
                Msc[lane + kFQ * 17] = xys;
                if constexpr (BIAS) Msc[lane + kFQ * 16] = css;
            }
            if (BIAS && lane == 0) Msc[16 + kFQ * 17] = yss;
        }
        if (sp.lambda > 0.0) {  // ridge: lambda goes onto the diagonal while the matrix is still in LDS
            PDS_WAVE_LDS_SYNC();
            if (lane < p) Msc[lane * (kFQ + 1)] += sp.lambda;  // (never on the intercept: lr_solvers.rs:199-208)
        }
        PDS_WAVE_LDS_SYNC();
        // ---- sub-group `npend` takes it into registers (solver layout: lane j = column j, a_p[i] = row i).
        // Rows / columns beyond p' read scratch slot ZSLOT (15; 7 in the packed kernels): whenever such a row exists
        // p <= ZSLOT, and that slot then holds the exact zeros the matrix core produced from the zero-padded tile
        // column -- so the 17 reads are
        // unconditional, under ONE exec mask (the taking sub-group), instead of a select per element.
        {
            int j = j_in;
            asm volatile("" : "+v"(j));
            const int lj = (j < p) ? j : ZSLOT;
            if (sub == npend) {
                const double* colp = Msc + kFQ * lj;
                if (p >= LPS) {  // every row is a feature: immediate offsets, the reads pair up into 16-byte loads
#pragma unroll
                    for (int i = 0; i < LPS; ++i) {
                        a_p[i] = colp[i];
                        if constexpr (!CHOL) b_p[i] = Msc[i + kFQ * 17];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < LPS; ++i) {
                        const int li = (i < p) ? i : ZSLOT;
                        a_p[i] = colp[li];
                        if constexpr (!CHOL) b_p[i] = Msc[li + kFQ * 17];
                    }
                }
                if constexpr (CHOL) a_p[LPS] = Msc[lj + kFQ * 17];
                dj_p = (j < pp) ? colp[lj] : 1.0;  // the gate's denominators are the uncentred diagonal entries
                null_p = too_few;  // per-group pl_lr raises "#Data < #features": reported as null
                if constexpr (BIAS) {  // centre: rows beyond p read the zero slot's column sum (0)
                    static_assert(CHOL, "the centred bias form exists for the Cholesky kernel");
                    const double csj = Msc[lj + kFQ * 16];
                    const double nn = (double)ng, sy = Msc[16 + kFQ * 17];
                    const double m = csj / nn;  // mean of the lane's feature
#pragma unroll
                    for (int i = 0; i < LPS; ++i) {
                        const int li = (p >= LPS || i < p) ? i : ZSLOT;
                        a_p[i] = fma(-Msc[li + kFQ * 16], m, a_p[i]);
                    }
                    a_p[LPS] = fma(-sy, m, a_p[LPS]);
                    cs_p = csj;
                    ys_p = sy;
                    n_p = nn;
                }
            }
        }
        PDS_WAVE_LDS_SYNC();
        zero_acc(acc);
        ++npend;
        PDS_T1(2);
        if (npend == SPW) solve_pending();
    };

    // ---- stream the tiles covering [rlo, rhi)
    TileRegs<T> regs;
    const int64_t t_first = rlo / TR, t_last = (rhi > rlo) ? (rhi - 1) / TR : t_first - 1;
    auto load_tile = [&](int64_t t) __attribute__((always_inline)) {
        // the 18-entry pointer table is re-read per tile (two wide scalar loads, one wait): keeping 17 base
        // pointers resident costs 34 SGPRs for the whole kernel and pushed the solver's scalars into spills
        ColPtrs<T> cp;
#pragma unroll
        for (int c = 0; c < 16; ++c) cp.x[c] = as_global(cols[c]);
        cp.y = as_global(cols[p]);
        cp.w = cp.y;
        const int64_t row = t * TR + lane * RPL;
        if ((t + 1) * TR <= n_rows) load_full_tile<T, false>(cp, p, row, regs);
        else load_tail_tile<T, false>(cp, p, row, n_rows, regs);
    };
    // (Round 2, measured on one box against 2.465 ms as is -- profiles/r02_grouped_variants_ab.txt, source kept as
    //  tools/experiments/grouped_fused_r02_spread_issue_and_half_tile.patch: the LDS tile holding HALF of the register tile at a
    //  time (11.3 KB per wave, three waves per SIMD) 2.61 ms; the next tile's 17 loads issued in four slices from inside the
    //  consume loop instead of one burst 4.3 ms.  The SIMD's issue slots -- matrix pipe + the solve's DPP arithmetic -- are the
    //  bound, not memory-level parallelism: DESIGN.md 4.2.)
    // (Tried and measured slower: a half-size tile (8 B per lane per column, 11.6 KB of LDS per wave) -- 3.47 ms per
    //  1e6-group step at 2 waves/SIMD and 3.59 ms squeezed into 168 VGPRs for 3 waves/SIMD, against 3.12 ms as is.)
    // (A flat state machine with flush / solve instantiated once each was tried: 4.7k instead of 10.9k static
    //  instructions but 255 live VGPRs and 8 % slower than this nested form at 219.)
    if (t_first <= t_last) load_tile(t_first);
    // One flush site: a group is flushed as soon as its last row has been consumed (the loop condition also holds while
    // the current group is complete), so groups that end exactly at rhi and trailing empty groups are handled by the last
    // pass; a wave whose groups are all empty makes one pass without a tile.  (Two inlined copies of flush + solve were
    // 5 KB of code for nothing.)
    for (int64_t t = t_first;; ++t) {
        const bool have_tile = t <= t_last;
        if (have_tile) {
            {
                PDS_T0();
                store_tile_lds<T, false>(wl, p, lane, regs);
#ifdef PDS_PROFILE_PHASES
                PDS_WAVE_LDS_SYNC();  // charge the LDS writes (and the wait for the tile) to this phase
#endif
                PDS_T1(0);
            }
            PDS_T0();
            if (t + 1 <= t_last) load_tile(t + 1);
            PDS_T1(1);
        }
        const int64_t row0 = t * TR;
        // rows of this tile that belong to the wave, relative to row0 (no tile: nothing to consume, flushes only)
        uint32_t rel_end = 0, pos_rel = 0;
        if (have_tile) {
            const uint64_t left = (uint64_t)(rhi - row0);
            rel_end = ((uint32_t)(left >> 32) != 0u || (uint32_t)left > (uint32_t)TR) ? (uint32_t)TR : (uint32_t)left;
            pos_rel = (uint32_t)(pos - row0);
        }
        // end of the current group relative to row0, saturated (ge >= pos >= row0 whenever a tile is present)
        auto rel_of = [&](int64_t r) __attribute__((always_inline)) {
            const uint64_t d = (uint64_t)(r - (have_tile ? row0 : pos));
            return ((uint32_t)(d >> 32) != 0u) ? 0xFFFFFFFFu : (uint32_t)d;
        };
        uint32_t ge_rel = rel_of(ge);
        while (g_left != 0u && (pos_rel < rel_end || ge_rel <= pos_rel)) {
            if (ge_rel <= pos_rel) {  // group complete (or empty)
                flush_group();
                ++g;
                --g_left;
                gs = ge;
                ge = ge_next;
                ge_next = offsets[g_left >= 2u ? g + 2 : gh];
                ge_rel = rel_of(ge);
                continue;
            }
            const uint32_t seg_end = (ge_rel < rel_end) ? ge_rel : rel_end;
            PDS_T0();
            if constexpr (PACK) consume_range_pack<T, BIAS>(wl, lane, (int)pos_rel, (int)seg_end, acc);
            else consume_range<T, BIAS>(wl, lane, (int)pos_rel, (int)seg_end, acc);
            PDS_T1(4);
            pos_rel = seg_end;
        }
        if (have_tile) pos = row0 + pos_rel;
        if (t >= t_last) break;
        PDS_WAVE_LDS_SYNC();
    }
    if (npend > 0) solve_pending();
#ifdef PDS_PROFILE_PHASES
    prof[7] = __builtin_amdgcn_s_memtime() - t_begin;
    if (lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_phase_cycles[k], prof[k]);
#endif
}

template <typename T, int LPS>
static int launch_stream_lps(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, const int64_t* d_offsets,
                             int64_t n_groups, int64_t n_rows, const SolveRegDev& sd, bool chol, T* d_coeffs,
                             uint8_t* d_flags, int32_t* d_mark_list, unsigned* d_mark_count, unsigned* d_mark_host) {
    const size_t lds = (size_t)kFWaveLds;
    const int per_cu = std::max(1, (int)((160 * 1024) / lds));
    int64_t nb = std::min<int64_t>(std::max<int64_t>(n_groups / (64 / LPS), 1), (int64_t)ctx->num_cus * per_cu);
    KernelTimer timer(ctx, kKindGroupedMoments);
    // (the pivoted-QR variant of this kernel measured slower than the two-kernel pipeline and is not instantiated;
    //  callers route the ungated case to grouped_moments_kernel + solve_reg_kernel)
    if (!chol) return fail(PDS_ERR_INVALID, "internal: fused grouped kernel is Cholesky-only");
    // the feature count is a compile-time constant of the kernel where that variant exists: every f64 count of this kernel
    // size, and p == LPS for f32 (a run-time count costs a scalar compare + branch per column and tile: 6 features ran
    // slower than 8, 3 slower than 4)
    auto go = [&](auto bias_c, auto pc_c) {
        hipLaunchKernelGGL((grouped_stream_kernel<T, LPS, true, decltype(bias_c)::value, decltype(pc_c)::value>), dim3((unsigned)nb),
                           dim3(64), lds, ctx->stream, dc.d_ptrs, n_feat, d_offsets, n_groups, n_rows, sd, d_coeffs, d_flags, d_mark_list,
                           d_mark_count, d_mark_host);
    };
    auto by_pc = [&](auto bias_c) {
        using std::integral_constant;
        constexpr int LO = LPS == 4 ? 1 : LPS / 2 + 1;  // feature counts served by this kernel size: LO .. LPS
        if (n_feat == LPS) return go(bias_c, integral_constant<int, LPS>{});
        if constexpr (sizeof(T) == 8) {
            if (n_feat == LO) return go(bias_c, integral_constant<int, LO>{});
            if constexpr (LPS >= 4) {
                if (n_feat == LO + 1 && LO + 1 < LPS) return go(bias_c, integral_constant<int, (LO + 1 < LPS ? LO + 1 : LPS)>{});
                if (n_feat == LO + 2 && LO + 2 < LPS) return go(bias_c, integral_constant<int, (LO + 2 < LPS ? LO + 2 : LPS)>{});
            }
            if constexpr (LPS == 16) {
                if (n_feat == 12) return go(bias_c, integral_constant<int, 12>{});
                if (n_feat == 13) return go(bias_c, integral_constant<int, 13>{});
                if (n_feat == 14) return go(bias_c, integral_constant<int, 14>{});
                if (n_feat == 15) return go(bias_c, integral_constant<int, 15>{});
            }
        }
        return go(bias_c, integral_constant<int, 0>{});
    };
    if (sd.bias) by_pc(std::true_type{});
    else by_pc(std::false_type{});
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

// ---- second pass: the groups the streaming kernel marked (flag 2) go through the reference's default factorisation
static int ensure_mark_state(pds_ctx* ctx) {
    if (ctx->mark_count) return PDS_OK;
    PDS_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx->mark_count), 256));
    // zeroed ON THE CONTEXT'S STREAM by the first launch (mark_dirty): a hipMemset here runs on the null stream, which a context's
    // own stream is not ordered behind -- the counter could be reset while the first kernel was already counting, and the
    // marked groups beyond the surviving count kept their provisional flag (seen as a flaky null-flag mismatch on fresh contexts)
    ctx->mark_dirty = true;
    PDS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ctx->mark_host), 64, hipHostMallocMapped | hipHostMallocCoherent));
    *ctx->mark_host = 0u;
    PDS_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->mark_host_dev), ctx->mark_host, 0));
    return PDS_OK;
}

template <typename T>
__global__ void scatter_marked_kernel(const T* __restrict__ co_c, const uint8_t* __restrict__ fl_c, const int32_t* __restrict__ list,
                                      int64_t n, int pp, T* __restrict__ coeffs, uint8_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * pp) return;
    const int64_t k = i / pp;
    const int c = (int)(i - k * pp);
    const int64_t g = list[k];
    coeffs[g * pp + c] = co_c[i];
    if (c == 0) flags[g] = fl_c[k] ? 1 : 0;
}

// OLS / ridge, p <= 16 features (+ intercept)
template <typename T>
int launch_grouped_fused(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, const int64_t* d_offsets,
                         int64_t n_groups, const SolveParams& sp, T* d_coeffs, uint8_t* d_flags, T* d_mom_scratch,
                         int64_t scratch_groups) {
    SolveRegDev sd;
    sd.p = sp.p;
    sd.bias = sp.add_bias ? 1 : 0;
    sd.pp = sp.p + sd.bias;
    sd.lambda_on_bias = sp.lambda_on_bias;
    sd.lambda = sp.lambda;
    sd.gate_on = sp.gate_tol > 0.0 ? 1 : 0;
    sd.ln_tol = sd.gate_on ? std::log(sp.gate_tol) : 0.0;
    sd.inv_tol = sd.gate_on ? 1.0 / sp.gate_tol : HUGE_VAL;
    if (n_groups <= 0) return PDS_OK;
    if (n_groups >= (1ll << 31)) return fail(PDS_ERR_INVALID, "internal: fused grouped kernel counts groups per wave in 32 bits");
    const char* piv = dev_env("PDS_GROUPED_PIVOTED");
    const bool chol = sd.gate_on && !(piv && piv[0] == '1');
    // solver = "choleskey" IS this kernel's factorisation (llt + the 2 sum ln L_ii gate, lr_solvers.rs:369-380); for "qr"
    // (default) and "svd" the kernel answers the clear cases and leaves the rest to the pivoted QR below
    const bool second_pass = chol && sp.solver != PDS_SOLVER_CHOLESKEY && d_flags && d_mom_scratch && scratch_groups > 0;
    sd.sus_tol = second_pass ? std::sqrt(sd.inv_tol) : 0.0;
    // (the intercept takes no solver lane: the kernel size follows the feature count)
    int32_t* d_list = nullptr;
    unsigned* d_count = nullptr;
    unsigned* d_host = nullptr;
    if (second_pass) {
        if (int rc0 = ensure_mark_state(ctx)) return rc0;
        d_list = reinterpret_cast<int32_t*>(ws_take(ctx, (size_t)n_groups * sizeof(int32_t)));
        if (!d_list) return fail(PDS_ERR_HIP, "workspace allocation failed");
        d_count = ctx->mark_count;
        d_host = ctx->mark_host_dev;
        *ctx->mark_host = 0u;  // (host-mapped word; the previous call synchronised before it returned)
        // The counter is back at zero after every call that ran to its end (below).  A call that failed between its launch and
        // that point leaves it wherever the kernel took it; the next launch would then append behind stale slots -- past the
        // end of its list.  So such a context re-zeroes the counter first (costs nothing on the common path).
        if (ctx->mark_dirty) PDS_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned), ctx->stream));
        ctx->mark_dirty = true;
    }
    int rc;
    if (sd.p <= 4) rc = launch_stream_lps<T, 4>(ctx, dc, n_feat, d_offsets, n_groups, n_rows, sd, chol, d_coeffs, d_flags, d_list, d_count, d_host);
    else if (sd.p <= 8) rc = launch_stream_lps<T, 8>(ctx, dc, n_feat, d_offsets, n_groups, n_rows, sd, chol, d_coeffs, d_flags, d_list, d_count, d_host);
    else rc = launch_stream_lps<T, 16>(ctx, dc, n_feat, d_offsets, n_groups, n_rows, sd, chol, d_coeffs, d_flags, d_list, d_count, d_host);
    if (rc || !second_pass) return rc;
    // ---- marked groups: Gram records (indexed grouped build) -> pivoted Householder QR with the log-det gate -> scatter.
    // Whether there are any is read from the host-mapped word after the kernel has finished (the entry points synchronise
    // before they return anyway); the common case -- none -- costs no launch, no copy and no counter read-back.
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (__atomic_load_n(ctx->mark_host, __ATOMIC_ACQUIRE) == 0u) {
        ctx->mark_dirty = false;  // no wave marked a group: the counter was never touched
        return PDS_OK;
    }
    unsigned h_count = 0;
    PDS_HIP_CHECK(hipMemcpyAsync(&h_count, d_count, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned), ctx->stream));  // ready for the next call
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->mark_dirty = false;
    const int64_t marked = (int64_t)h_count;
    if (marked == 0) return PDS_OK;
    const int pp = sd.pp;
    const int64_t chunk = std::min<int64_t>(scratch_groups, marked);
    T* co_c = reinterpret_cast<T*>(ws_take(ctx, (size_t)chunk * pp * sizeof(T)));
    uint8_t* fl_c = reinterpret_cast<uint8_t*>(ws_take(ctx, (size_t)chunk));
    if (!co_c || !fl_c) return fail(PDS_ERR_HIP, "workspace allocation failed");
    for (int64_t k0 = 0; k0 < marked; k0 += chunk) {
        const int64_t kc = std::min(chunk, marked - k0);
        if (int rc2 = launch_grouped_moments<T>(ctx, dc, n_feat, d_offsets, kc, d_mom_scratch, d_list + k0)) return rc2;
        if (int rc2 = launch_solve_marked<T>(ctx, d_mom_scratch, kc, sp, co_c, fl_c, nullptr)) return rc2;  // (pivoted QR; "svd": SVD gate)
        hipLaunchKernelGGL((scatter_marked_kernel<T>), dim3((unsigned)((kc * pp + 255) / 256)), dim3(256), 0, ctx->stream, co_c, fl_c,
                           d_list + k0, kc, pp, d_coeffs, d_flags);
        PDS_HIP_CHECK(hipGetLastError());
    }
    return PDS_OK;
}

#ifdef PDS_PROFILE_PHASES
extern "C" int pds_debug_phase_cycles(unsigned long long* out, int reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), sizeof(z)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

template int launch_grouped_fused<double>(pds_ctx*, const DeviceCols<double>&, int, int64_t, const int64_t*, int64_t,
                                          const SolveParams&, double*, uint8_t*, double*, int64_t);
template int launch_grouped_fused<float>(pds_ctx*, const DeviceCols<float>&, int, int64_t, const int64_t*, int64_t,
                                         const SolveParams&, float*, uint8_t*, float*, int64_t);

}  // namespace pds
