// rolling_pair_dev.hpp -- rolling / expanding fits, f64 frames, p' in {2, 4, 6, 8}: TWO LANES PER CHAIN OF ROWS, two waves per SIMD.
//
// rolling_seg_dev.hpp (lane = 4 consecutive rows) keeps 46 running moments, a 36-entry work copy and 8 rows in one lane:
// ~380 registers, ONE wave per SIMD, and every dependent f64 operation of the per-row factorisation exposes its latency
// (0.47 - 0.49 of the HBM roofline at C4 for two rounds).  Here the running moments of a chain of rows are SPLIT
// over a pair of lanes and each lane factors every other row, so a lane holds half the state:
//   * lanes (2c, 2c+1) of a 16-lane DPP row = chain c (8 chains per row, K = 4 consecutive rows per chain and stage); a DPP row
//     is a SUB-STREAM with its own tile of the frame: a wave walks four tiles in lockstep, 32 rows of each per stage;
//   * the even lane labels the p' variables naturally, the odd lane REVERSED (pi(a) = p'-1-a).  A lane keeps the running
//     moments z_a z_b of the label pairs with a + b <= p'-1 only (NA = p'^2/4 + p'/2 of the p'(p'+1)/2) plus z_a y for
//     a < p'/2 and the finite-row count: in true indices the two lanes hold complementary halves (the p'/2 pairs {a, pi(a)} are
//     held by both), and BOTH RUN THE SAME INSTRUCTIONS -- only the LDS column a label reads differs (a per-lane address);
//   * pass 1 sums the chain's increments m(r) - m(r - w) of its K rows, a rotate-and-scan over the 8 chains of the row
//     (row_ror:2 brings the previous stage's end state of chain 7 in front, row_shr:2/4/8 are the inclusive scan: the shifts
//     are even, so the two parities never mix) turns them into every chain's state in front of its first row -- no LDS,
//     no wave-wide scan;
//   * pass 2 walks the rows in pairs: Q = P + m(2m), P = Q + m(2m+1) (the alternation keeps BOTH states without a copy), the even
//     lane then solves row 2m and the odd lane row 2m+1 in the same instruction stream: its work copy is its own half (a select
//     between Q and P by parity) and the partner's half of the SAME state (the partner's opposite select through
//     quad_perm [1,0,3,2]); square-root-free L D L' in place, two substitutions, pred, stores (the odd lane writes its
//     coefficients back in true order);
//   * the rows live in LDS, not in registers: one image of the stage's rows and one of the rows leaving the window (1 KiB per
//     column and image, fetched global -> LDS directly, 16 bytes per lane), read back 16 bytes (rows 2m, 2m+1) at a time.
// ~200 registers and 18 KB of LDS per wave: two waves per SIMD, eight per CU; the second wave covers the dependent chains and
// the (synchronous) stage fetch of the first.  tools/models/rolling_pair_model.py is a lane-level NumPy model of exactly this data flow
// (slot tables, parity permutation, rotate-and-scan, P / Q alternation) checked against direct window solves.
//
// Semantics as rolling_seg_kernel (lr_online_solvers.rs:85-89, 148-301; linear_regression.rs:1121-1283): non-finite rows are
// left out of the sums and counted out of the window, lambda on every diagonal, tiles anchored exactly.
#pragma once
#include "common.hpp"
#include "solve_reg_dev.hpp"

namespace pds {

constexpr int kPairK = 4;          // rows per chain and stage
constexpr int kPairSub = 32;       // rows per sub-stream and stage (8 chains x K)
constexpr int kPairTile = 4096;    // rows per sub-stream tile (== kSegTile / kTileRows: the expanding fit's tile totals are shared)

template <int PP>
struct PairDims {
    static_assert(PP % 2 == 0 && PP >= 2 && PP <= 8, "an even number of coefficients up to 8");
    static constexpr int H = PP / 2;
    static constexpr int NA = PP * PP / 4 + PP / 2;  // own Gram entries: label pairs (a <= b, a + b <= PP - 1)
    static constexpr int NS = NA + H + 1;            // + own z_a y (a < H) + the finite-row count
    static constexpr int NG = PP * (PP + 1) / 2;
    static constexpr int NV = NG + PP + 1;           // rolling_seg_kernel's moment vector (tile totals of the expanding fit)
    static constexpr int IMG = (PP + 1) * 1024;      // bytes of one LDS image: PP variable slots + y, 1 KiB each
    static constexpr int LDS_BYTES = 2 * IMG;
    static __host__ __device__ constexpr int slot(int a, int b) { return a * PP - a * (a - 1) + (b - a); }   // own pair -> slot
    static __host__ __device__ constexpr int lin(int a, int b) { return a * PP - (a * (a - 1)) / 2 + (b - a); }  // upper triangle, row-major
};

typedef double pair_d2 __attribute__((ext_vector_type(2)));

// A phase boundary: neither memory accesses nor arithmetic move across it.  Left alone, the scheduler overlaps one pair of
// rows' accumulation with the previous pair's factorisation (they are independent) and hoists the next rows' LDS reads -- good for
// one wave, but the live ranges then overflow the 256 registers that two waves per SIMD leave (37 - 50 spilled registers, reloaded
// inside the stage loop behind vmcnt(0) waits).  The second wave of the SIMD is what overlaps the phases here.
#define PAIR_PHASE()                          \
    do {                                      \
        asm volatile("" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

template <int PP, int MODE, int FULLP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void rolling_pair_kernel(
    const double* const* __restrict__ cols, RollArgs ra_in, const double* __restrict__ tile_tot, double* __restrict__ coeffs,
    double* __restrict__ pred, uint8_t* __restrict__ valid) {
    using PD = PairDims<PP>;
    static_assert(MODE == 0 || MODE == 2, "rolling or the main pass of the expanding fit");
    static_assert(FULLP == 1 || FULLP == 2, "p' == PP: PP features, or PP - 1 features and the bias column");
    constexpr int H = PD::H, NA = PD::NA, NS = PD::NS, NG = PD::NG, NV = PD::NV, IMG = PD::IMG;
    constexpr int P_FEAT = FULLP == 1 ? PP : PP - 1;  // feature columns; column P_FEAT of the table is y
    constexpr int NWHICH = MODE == 0 ? 2 : 1;
    constexpr int kBig = 1 << 30;
    extern __shared__ __attribute__((aligned(16))) double pair_lds[];
    typedef __attribute__((address_space(3))) char* lds_c;
    typedef __attribute__((address_space(3))) void* lds_v;
    typedef const __attribute__((address_space(1))) void* glb_v;
    typedef __attribute__((address_space(3))) pair_d2* lds_d2;
    typedef __attribute__((address_space(3))) double* lds_d;
    lds_c sm = (lds_c)reinterpret_cast<char*>(pair_lds);

    const int lane = threadIdx.x & 63;
    const bool odd = lane & 1;
    const int ch = (lane >> 1) & 7;   // chain inside the 16-lane row
    const int R = lane >> 4;          // sub-stream of the wave
    // LDS slot s (16 bytes = 2 consecutive rows of a column) = h * 32 + (R >> 1) * 16 + (R & 1) * 8 + ch holds rows 4 ch + 2 h,
    // + 1 of sub-stream R: the 16 (R & 1, ch) readers of a 32-lane group land on 16 different bank pairs.
    // loader lane L fills slot L:
    const int hL = lane >> 5, RL = ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1), chL = lane & 7;
    // reader: row i of my chain sits at LB + (i >> 1) * 512 + (i & 1) * 8 of a 1 KiB column piece; label a reads the piece of
    // variable a (even lane) or PP - 1 - a (odd lane): piece offset = colB + a * colS
    const int LB = (R >> 1) * 256 + (R & 1) * 128 + ch * 16;
    const int colB = LB + (odd ? (PP - 1) * 1024 : 0), colS = odd ? -1024 : 1024;
    const int colY = LB + PP * 1024;

    const int64_t n = ra_in.n, w = ra_in.window;
    const int64_t min_size = ra_in.min_size;
    const double lambda = ra_in.lambda;
    const int64_t ntiles = (n + kPairTile - 1) / kPairTile;

    gptr<double> cp[P_FEAT + 1];
#pragma unroll
    for (int c = 0; c <= P_FEAT; ++c) cp[c] = as_global(cols[c]);

    // Row numbers inside a round are 32-bit offsets from the round's first row rb (a wave's four tiles are consecutive):
    //   nrel = rows of the frame from rb on, wrel = first offset whose row has a full window behind it (both clamped)
    int64_t rb = 0;
    int nrel = 0, wrel = 0;
    // ---- one stage's rows of sub-stream RL: global -> LDS (the loader view of the lane)
    auto load_stage = [&](int st) __attribute__((always_inline)) {
        if constexpr (FULLP == 2) {  // the bias variable's slot: ones (no load lands there; pass 1 zeroes rows that do not count)
            pair_d2 one2 = {1.0, 1.0};
#pragma unroll
            for (int wh = 0; wh < NWHICH; ++wh) *(lds_d2)(sm + wh * IMG + (PP - 1) * 1024 + lane * 16) = one2;
        }
        const int t0L = RL * kPairTile;
        const bool tvalidL = t0L < nrel;
        // first of the lane's two rows; a lane of an absent tile fetches rows nobody reads from a place that is inside the frame
        // whenever the other lanes' pieces are
        const int g = tvalidL ? t0L + st * kPairSub + 4 * chL + 2 * hL : (MODE == 0 ? wrel : 0);
        bool inside = g + 2 <= nrel;
        if constexpr (MODE == 0) inside = inside && (g >= wrel);
        // addresses = wave-uniform base (column + round) + ONE 32-bit lane offset: no per-column 64-bit lane pointers for the
        // loop optimiser to keep alive across the stage loop (18 of them were 36 registers -- the spills)
        const unsigned goff = (unsigned)g * 8u;
        typedef const __attribute__((address_space(1))) char* glb_c;
        if (__builtin_amdgcn_ballot_w64(!inside) == 0) {
            // every piece of the wave inside the frame: asynchronous 1 KiB bursts (global_load_lds_dwordx4)
#pragma unroll
            for (int wh = 0; wh < NWHICH; ++wh) {
#ifdef PDS_PAIR_EXP_NO_OLD_DMA
                if (wh == 1) continue;  // (timing experiment: what the kernel would cost if the leaving rows were already in LDS)
#endif
#pragma unroll
                for (int c = 0; c <= P_FEAT; ++c) {
                    const int slotc = c < P_FEAT ? c : PP;
                    glb_c base = (glb_c)(cp[c] + (wh == 0 ? rb : rb - w));
                    __builtin_amdgcn_global_load_lds((glb_v)(base + goff), (lds_v)(sm + wh * IMG + slotc * 1024), 16, 0, 0);
                }
            }
        } else {
            // a frame edge (first / last stages of the frame, the first w rows): guarded loads, committed by hand
#pragma unroll
            for (int wh = 0; wh < NWHICH; ++wh) {
                const int64_t r = rb + g - (wh == 0 ? 0 : w);
                const bool in0 = tvalidL && r >= 0 && r < n, in1 = tvalidL && r + 1 >= 0 && r + 1 < n;
                const int64_t r0c = in0 ? r : 0, r1c = in1 ? r + 1 : 0;
#pragma unroll
                for (int c = 0; c <= P_FEAT; ++c) {
                    const int slotc = c < P_FEAT ? c : PP;
                    // (unconditional loads from clamped rows: a load under its own exec mask is serialised behind the previous one)
                    const double a0 = cp[c][r0c], a1 = cp[c][r1c];
                    pair_d2 v;
                    v.x = in0 ? a0 : 0.0;
                    v.y = in1 ? a1 : 0.0;
                    *(lds_d2)(sm + wh * IMG + slotc * 1024 + lane * 16) = v;
                }
            }
        }
    };
    // rows (2 m, 2 m + 1) of my chain out of image `wh`, label order (pass 1)
    auto read_rows2 = [&](int wh, int m, double (&z0)[PP], double& y0, double (&z1)[PP], double& y1) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < PP; ++a) {
            const pair_d2 v = *(const lds_d2)(sm + (colB + a * colS) + wh * IMG + m * 512);
            z0[a] = v.x;
            z1[a] = v.y;
        }
        const pair_d2 v = *(const lds_d2)(sm + colY + wh * IMG + m * 512);
        y0 = v.x;
        y1 = v.y;
    };
    // row i of my chain (pass 2: one row at a time keeps the live rows at two)
    auto read_row = [&](int wh, int i, double (&z)[PP], double& y) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < PP; ++a) z[a] = *(const lds_d)(sm + (colB + a * colS) + wh * IMG + (i >> 1) * 512 + (i & 1) * 8);
        y = *(const lds_d)(sm + colY + wh * IMG + (i >> 1) * 512 + (i & 1) * 8);
    };
    auto finite_row = [&](const double (&z)[PP], double y) __attribute__((always_inline)) {
        bool fin = isfinite(y);
#pragma unroll
        for (int a = 0; a < PP; ++a) fin = fin && isfinite(z[a]);
        return fin;
    };
    auto zero_row = [&](double (&z)[PP], double& y) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < PP; ++a) z[a] = 0.0;
        y = 0.0;
    };
    // dst = src + sign * m(row) over the lane's own slots (a row that does not count arrives zeroed; dcnt: the change of the
    // finite-row count).  The sign rides on the FMA's source modifier.
    auto advance = [&](double (&dst)[NS], const double (&src)[NS], const double (&z)[PP], double y, double sign,
                       double dcnt) __attribute__((always_inline)) {
        int k = 0;
#pragma unroll
        for (int a = 0; a < H; ++a) {
            const double za = sign > 0.0 ? z[a] : -z[a];
#pragma unroll
            for (int b = a; b <= PP - 1 - a; ++b) {
                dst[k] = fma(za, z[b], src[k]);
                ++k;
            }
        }
#pragma unroll
        for (int a = 0; a < H; ++a) dst[NA + a] = fma(sign > 0.0 ? z[a] : -z[a], y, src[NA + a]);
        dst[NS - 1] = src[NS - 1] + dcnt;
    };

#ifdef PDS_PROFILE_ROLLING
    unsigned long long rprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    double P[NS], Q[NS];  // state after the odd / even rows of the chain (pass 2); pass 1: Q = the chain's own increments
    const int64_t round_tiles = (int64_t)gridDim.x * 4;
    for (int64_t tb = (int64_t)blockIdx.x * 4; tb < ntiles; tb += round_tiles) {
        rb = tb * kPairTile;
        nrel = (n - rb < kBig) ? (int)(n - rb) : kBig;
        wrel = (w - rb > 0) ? ((w - rb < kBig) ? (int)(w - rb) : kBig) : 0;
        const int t0 = R * kPairTile;
        const bool tvalid = t0 < nrel;
        const int t1 = tvalid ? ((t0 + kPairTile < nrel) ? t0 + kPairTile : nrel) : 0;  // (no row of an absent tile counts)
        const int nst = ((nrel < kPairTile ? nrel : kPairTile) + kPairSub - 1) / kPairSub;  // stages of the round's longest tile
        RT0();
        load_stage(0);
        // ---- anchor: every lane's half of the state in front of the tile
        if constexpr (MODE == 2) {
            // exclusive prefix over the tiles' totals (rolling_totals_kernel + tile prefix), rolling_seg_kernel's moment order
            const double* tt = tile_tot + (tvalid ? (tb + R) * NV : 0);
            // (the parity passes through an opaque register here: hoisted out of the round loop, the 25 parity-selected indices
            //  below stay alive across the whole kernel -- 37 spilled registers)
            int oddv = odd ? 1 : 0;
            asm volatile("" : "+v"(oddv));
            const bool odd = oddv != 0;
            int k = 0;
#pragma unroll
            for (int a = 0; a < H; ++a) {
#pragma unroll
                for (int b = a; b <= PP - 1 - a; ++b) {
                    const double v = tt[odd ? PD::lin(PP - 1 - b, PP - 1 - a) : PD::lin(a, b)];
                    P[k] = tvalid ? v : 0.0;
                    ++k;
                }
            }
#pragma unroll
            for (int a = 0; a < H; ++a) {
                const double v = tt[NG + (odd ? PP - 1 - a : a)];
                P[NA + a] = tvalid ? v : 0.0;
            }
            {
                const double v = tt[NV - 1];
                P[NS - 1] = tvalid ? v : 0.0;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NS; ++k) P[k] = 0.0;
            // chain c sums rows t0 - w + c, + 8, ... (both lanes of the pair read the same row, each in its own column order)
            const int64_t a0 = rb + t0 - w;
#pragma unroll 2
            for (int64_t it = 0; it < w; it += 8) {
                const int64_t r = a0 + it + ch;
                const bool in = tvalid && r >= 0 && (it + ch < w);
                const int64_t rc = in ? r : 0;
                double z[PP], y;
#pragma unroll
                for (int a = 0; a < PP; ++a) {
                    // unconditional loads from a clamped row
                    const int ce = a < P_FEAT ? a : 0, co = PP - 1 - a < P_FEAT ? PP - 1 - a : 0;
                    const double xe = (a < P_FEAT) ? cp[ce][rc] : 1.0;
                    const double xo = (PP - 1 - a < P_FEAT) ? cp[co][rc] : 1.0;
                    z[a] = odd ? xo : xe;
                }
                y = cp[P_FEAT][rc];
                const bool ok = in && finite_row(z, y);
                if (!ok) zero_row(z, y);
                advance(P, P, z, y, 1.0, ok ? 1.0 : 0.0);
            }
            // total over the 8 chains of the row; the butterflies are even shifts: each lane ends with its parity's total
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                double v = P[k];
                v += dpp_mov<kXor2>(0.0, v);
                v += dpp_mov<kRor4>(0.0, v);
                v += dpp_mov<kRor8>(0.0, v);
                P[k] = v;
            }
        }
        RT1(0);  // anchor
        for (int st = 0; st < nst; ++st) {
            const int r0 = t0 + st * kPairSub + 4 * ch;  // first row of my chain (offset from rb)
            RTA();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stage image has landed (the compiler does not order LDS reads behind global_load_lds)
            PDS_WAVE_LDS_SYNC();
            RT1(1);  // waiting for the stage image
            RTA();
            // ---- pass 1: the chain's own increments; which rows count.  A row that does not count (outside the tile, in front of
            // the frame, holding a non-finite value: OnlineLR::update lr_online_solvers.rs:85-89) is skipped by an exec-mask branch
            // around its accumulation and ZEROED IN THE IMAGE, so that pass 2 is branch free.
            unsigned flags = 0;  // bit i: row i of the chain counts; bit 4 + i: the row leaving the window with it counts
#pragma unroll
            for (int k = 0; k < NS; ++k) Q[k] = 0.0;
#pragma unroll
            for (int wh = 0; wh < NWHICH; ++wh) {
#pragma unroll
                for (int m = 0; m < kPairK / 2; ++m) {
                    // two rows per 16-byte read; one pair of rows live at a time (the phase marks keep the reads behind the previous
                    // pair's work -- hoisted, they double the live rows)
                    PAIR_PHASE();
                    double z[2][PP], y[2];
                    read_rows2(wh, m, z[0], y[0], z[1], y[1]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int r = r0 + 2 * m + j;
                        const bool ok = (r < t1) && (wh == 0 || r >= wrel) && finite_row(z[j], y[j]);
                        flags |= (ok ? 1u : 0u) << (4 * wh + 2 * m + j);
                        if (ok) {
                            advance(Q, Q, z[j], y[j], wh == 0 ? 1.0 : -1.0, wh == 0 ? 1.0 : -1.0);
                        } else {
#pragma unroll
                            for (int a = 0; a < PP; ++a) *(lds_d)(sm + (colB + a * colS) + wh * IMG + m * 512 + j * 8) = 0.0;
                            *(lds_d)(sm + colY + wh * IMG + m * 512 + j * 8) = 0.0;
                        }
                    }
                }
            }
            PDS_WAVE_LDS_SYNC();  // (the zeroed rows are read back by pass 2, by both lanes of the pair)
#ifdef PDS_PROFILE_ROLLING
            asm volatile("" ::"v"(Q[0]), "v"(Q[NS - 1]));
#endif
            RT1(2);  // pass 1
            RTA();
            // ---- rotate and scan over the 8 chains: P <- state in front of the chain's first row
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                double v = (ch == 7) ? P[k] : Q[k];   // chain 7 contributes the end state of the previous stage (tile: the anchor)
                v = dpp_mov<kRor2>(0.0, v);            // ... which goes in front: [end, S_0, .., S_6]
                v += dpp_mov<0x112 /*row_shr:2*/>(0.0, v);
                v += dpp_mov<0x114 /*row_shr:4*/>(0.0, v);
                v += dpp_mov<0x118 /*row_shr:8*/>(0.0, v);
                P[k] = v;
            }
#ifdef PDS_PROFILE_ROLLING
            asm volatile("" ::"v"(P[0]), "v"(P[NS - 1]));
#endif
            RT1(3);  // rotate and scan
            RTA();
            // ---- pass 2: two rows at a time
#pragma unroll
            for (int m = 0; m < kPairK / 2; ++m) {
                double zs[PP];  // my row's variables (pred): row 2 m in the even lane, 2 m + 1 in the odd lane
                {
                    PAIR_PHASE();
                    double z[PP], y;
                    read_row(0, 2 * m, z, y);
                    advance(Q, P, z, y, 1.0, (double)((flags >> (2 * m)) & 1u));  // state after row 2 m ...
#pragma unroll
                    for (int a = 0; a < PP; ++a) zs[a] = z[a];
                    if constexpr (MODE == 0) {
                        read_row(1, 2 * m, z, y);
                        advance(Q, Q, z, y, -1.0, -(double)((flags >> (4 + 2 * m)) & 1u));  // ... without the row that left
                    }
                }
                {
                    PAIR_PHASE();
                    double z[PP], y, zo[PP], yo;
                    read_row(0, 2 * m + 1, z, y);
                    if constexpr (MODE == 0) read_row(1, 2 * m + 1, zo, yo);
                    if (m == kPairK / 2 - 1) {
                        // the images are dead from here: the next stage's rows go in flight behind the last pair's arithmetic
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        if (st + 1 < nst) load_stage(st + 1);
                    }
                    advance(P, Q, z, y, 1.0, (double)((flags >> (2 * m + 1)) & 1u));  // state after row 2 m + 1
#pragma unroll
                    for (int a = 0; a < PP; ++a) zs[a] = odd ? z[a] : zs[a];
                    if constexpr (MODE == 0) advance(P, P, zo, yo, -1.0, -(double)((flags >> (5 + 2 * m)) & 1u));
                }
                // ---- the work copy of MY row (even lane: row 2 m, state Q; odd lane: row 2 m + 1, state P), label order
                PAIR_PHASE();
                double g[NG], c[PP];
                {
                    int k = 0;
#pragma unroll
                    for (int a = 0; a < H; ++a) {
#pragma unroll
                        for (int b = a; b <= PP - 1 - a; ++b) {
                            g[PD::lin(a, b)] = odd ? P[k] : Q[k];
                            if (a + b < PP - 1) {
                                // what the partner is missing of ITS state: my half of it (its labels (PP-1-b, PP-1-a))
                                const double send = odd ? Q[k] : P[k];
                                g[PD::lin(PP - 1 - b, PP - 1 - a)] = dpp_mov<kXor1>(0.0, send);
                            }
                            ++k;
                        }
                    }
#pragma unroll
                    for (int a = 0; a < H; ++a) {
                        c[a] = odd ? P[NA + a] : Q[NA + a];
                        const double send = odd ? Q[NA + a] : P[NA + a];
                        c[PP - 1 - a] = dpp_mov<kXor1>(0.0, send);
                    }
                }
                const double cnt = odd ? P[NS - 1] : Q[NS - 1];
                const bool okrow = (flags >> (2 * m + (odd ? 1 : 0))) & 1u;
#define GI(a, b) g[PD::lin(a, b)]
                PAIR_PHASE();
                // ---- square-root-free L D L' of (G + lambda I), in place (upper triangle; row k = l_.k and 1 / d_k after step k)
                bool okc = true;
#pragma unroll
                for (int k = 0; k < PP; ++k) {
                    const double d = GI(k, k) + lambda;
                    okc = okc && (d > 0.0);
                    double x = __builtin_amdgcn_rcp(d);
#pragma unroll
                    for (int it = 0; it < PDS_RCP_NEWTON; ++it) x = x * fma(-d, x, 2.0);
#pragma unroll
                    for (int a = k + 1; a < PP; ++a) {
                        const double tka = GI(k, a) * x;  // l_ak
#pragma unroll
                        for (int b = a; b < PP; ++b) GI(a, b) = fma(-tka, GI(k, b), GI(a, b));
                        GI(k, a) = tka;
                    }
                    GI(k, k) = x;
                }
                // L u = c, D v = u, L' beta = v
#pragma unroll
                for (int a = 1; a < PP; ++a) {
#pragma unroll
                    for (int k = 0; k < a; ++k) c[a] = fma(-GI(k, a), c[k], c[a]);
                }
#pragma unroll
                for (int a = 0; a < PP; ++a) c[a] *= GI(a, a);
#pragma unroll
                for (int a = PP - 2; a >= 0; --a) {
#pragma unroll
                    for (int k = a + 1; k < PP; ++k) c[a] = fma(-GI(a, k), c[k], c[a]);
                }
#undef GI
                // ---- pred and stores of my row
                PAIR_PHASE();
                const int r = r0 + 2 * m + (odd ? 1 : 0);
                if (r < t1) {
                    const double nanv = __builtin_nan("");
                    bool v_ok = r + 1 >= wrel;  // row >= w - 1
                    if (min_size > 0) v_ok = v_ok && (cnt >= (double)min_size);
                    double pr = 0.0;
#pragma unroll
                    for (int a = 0; a < PP; ++a) pr = fma(zs[a], c[a], pr);
                    const bool good = v_ok && okc;
                    // (wave-uniform base + 32-bit lane offset, as in load_stage)
                    char* outc = reinterpret_cast<char*>(coeffs + rb * (int64_t)PP) + (unsigned)r * (unsigned)(PP * 8);
#pragma unroll
                    for (int j = 0; j < PP; j += 2) {
                        // true positions j, j + 1: the odd lane's labels run backwards
                        pair_d2 o;
                        o.x = good ? (odd ? c[PP - 1 - j] : c[j]) : nanv;
                        o.y = good ? (odd ? c[PP - 2 - j] : c[j + 1]) : nanv;
                        *reinterpret_cast<d2u*>(outc + j * 8) = o;
                    }
                    *reinterpret_cast<double*>(reinterpret_cast<char*>(pred + rb) + (unsigned)r * 8u) =
                        (good && okrow) ? pr : nanv;  // (a non-finite row: x_r . beta is NaN in the reference too)
                    (valid + rb)[(unsigned)r] = v_ok ? 1 : 0;
                }
#ifdef PDS_PROFILE_ROLLING
                if (m == 0) {
                    RT1(4);  // pass 2, rows 0 - 1
                    RTA();
                }
#endif
            }
            RT1(5);  // pass 2, rows 2 - 3 (+ the next stage's loads issued)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PDS_WAVE_LDS_SYNC();  // the next round's first stage overwrites the images
    }
#ifdef PDS_PROFILE_ROLLING
    rprof[7] = __builtin_amdgcn_s_memtime() - t_begin;
    if (lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_roll_cycles[k], rprof[k]);
#endif
}

}  // namespace pds
