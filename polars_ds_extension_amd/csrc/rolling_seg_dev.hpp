// rolling_seg_dev.hpp -- rolling / expanding fits with up to 8 coefficients: lane = K CONSECUTIVE rows.
//
// rolling.hip's first kernel makes lane = one row of a 64-row step and sends the NV = p'(p'+1)/2 + p' + 1 running moments
// on a round trip through LDS every step (increments out, a scan by lane = moment, window sums back): ~270 LDS
// instructions of ~1000 per 64 rows, at two waves per SIMD -- an LDS- and issue-bound step at 0.33 of the HBM roofline.
// Here a lane owns K = 4 consecutive rows of a 256-row stage and the running moments stay in its registers:
//   pass 1   S_l  = sum over the lane's K rows of  m(r) - m(r - w)                       2 NV FMAs per row
//   scan     P_l  = carry + exclusive prefix of S over the lanes -- ONE trip through LDS per STAGE: 45 values out, lane =
//                   moment runs the 64-long prefix in place, 45 values back (the round trip is amortised over K rows)
//   pass 2   S = P_l; per row: S += m(r) - m(r - w); L D L' of (S + lambda I) in a work copy; beta; pred; stores
// with m(r) = [upper triangle of z z', z y, 1] of row r (zeros for a non-finite row: OnlineLR::update lr_online_solvers.rs:85-89).
// The lane's K rows of a column are K * 8 contiguous bytes and lanes are contiguous, so a stage reads whole lines and a lane
// writes its K coefficient rows as K * p' * 8 contiguous bytes.
// One wave per SIMD (the running sums, the work copy and 2 K rows need ~400 registers); what hides the HBM latency is a
// register-staged prefetch THROUGH LDS: while pass 2 computes, the next stage's rows are loaded 16 bytes per lane (fully
// coalesced 1 KiB pieces, a quarter of them in front of every row of pass 2) and parked in the wave's LDS region, from
// where the lanes pick up their own K rows at the start of the next stage.  The same region carries the scan (the rows are
// in registers by then): 36 KB per wave, four waves per CU.
// Tiles of kSegTile rows are anchored exactly as in rolling.hip (window in front of the tile summed cooperatively; expanding:
// exclusive prefix over per-tile totals), so round-off never accumulates over more than one tile.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "moments_dev.hpp"

// The next stage's rows go global -> LDS directly (below); -DPDS_ROLL_NO_LDS_DIRECT keeps the register-staged prefetch (A/B:
// 5.31 -> 4.54 ms at C4, profiles/r02_rolling_variants_ab.txt)
#ifndef PDS_ROLL_NO_LDS_DIRECT
#define PDS_ROLL_LDS_DIRECT 1
#endif

namespace pds {

constexpr int kSegK = 4;                  // consecutive rows per lane
constexpr int kSegStage = 64 * kSegK;     // rows per stage
constexpr int kSegTile = 4096;            // rows per tile (== rolling.hip's kTileRows: the expanding pass shares its tile totals)
constexpr int kSegStride = 65;            // doubles per moment row of the scan area

template <typename T, int PP, int OLDREG = 0>
struct SegDims {
    static constexpr int NG = PP * (PP + 1) / 2;
    static constexpr int NV = NG + PP + 1;
    static constexpr int E16 = 16 / (int)sizeof(T);                 // elements per 16-byte lane load
    static constexpr int PIECES = kSegK / E16;                      // 1 KiB pieces per stream and stage
    static constexpr int STREAM_BYTES = kSegStage * (int)sizeof(T);
    static constexpr int NSTREAM = (OLDREG ? 1 : 2) * (PP + 1);     // new (+ old) rows of [features.., y]
    static constexpr int STAGE_BYTES = NSTREAM * STREAM_BYTES;
    static constexpr int SCAN_BYTES = NV * kSegStride * 8;
    static constexpr int LDS_BYTES = STAGE_BYTES > SCAN_BYTES ? STAGE_BYTES : SCAN_BYTES;
};

// rows per tile of the ROLLING form (the expanding form shares kSegTile with rolling.hip's totals): the anchor of a tile reads
// the window in front of it once more and the first stage of a tile waits for its rows, so longer tiles amortise both --
// as long as there are enough tiles to balance the waves
// Newton steps x <- x (2 - d x) behind v_rcp_f64 for a pivot reciprocal (tools/rcp_accuracy.hip measures what the instruction
// delivers on its own and after each step)
#ifndef PDS_RCP_NEWTON
#define PDS_RCP_NEWTON 2
#endif
#ifndef PDS_ROLL_TILE
#define PDS_ROLL_TILE 16384
#endif
constexpr int kSegTileRoll = PDS_ROLL_TILE;
static_assert(kSegTileRoll % kSegStage == 0, "tiles are whole stages");

template <int PP>
struct SegRow {
    double z[PP], y;
};

// S += m(row) (SIGN = +1) or S -= m(row): Gram upper triangle (a <= b, row-major), then z y, then the finite-row count.
// (the negation rides on the FMA's source modifier: no extra instruction)
template <int PP, int NV, int SIGN>
__device__ __forceinline__ void seg_accumulate(double (&S)[NV], const SegRow<PP>& r, bool ok) {
    int v = 0;
#pragma unroll
    for (int a = 0; a < PP; ++a) {
        const double sa = SIGN > 0 ? r.z[a] : -r.z[a];
#pragma unroll
        for (int b = a; b < PP; ++b) {
            S[v] = fma(sa, r.z[b], S[v]);
            ++v;
        }
    }
#pragma unroll
    for (int a = 0; a < PP; ++a) {
        S[v] = fma(SIGN > 0 ? r.z[a] : -r.z[a], r.y, S[v]);
        ++v;
    }
    S[v] += ok ? (double)SIGN : 0.0;
}

template <int PP>
__device__ __forceinline__ bool seg_finite(const SegRow<PP>& r) {
    bool fin = isfinite(r.y);
#pragma unroll
    for (int a = 0; a < PP; ++a) fin = fin && isfinite(r.z[a]);
    return fin;
}
template <int PP>
__device__ __forceinline__ void seg_zero(SegRow<PP>& r) {
#pragma unroll
    for (int a = 0; a < PP; ++a) r.z[a] = 0.0;
    r.y = 0.0;
}

// MODE 0: rolling window.  MODE 2: expanding, main pass (tile_tot holds the exclusive prefix over the tiles' totals, written
// by rolling.hip's totals pass + tile_prefix_kernel).  FULLP as in rolling.hip (1: p == PP, no bias; 2: p == PP - 1 + bias).
// OLDREG (MODE 0, window == kSegStage): the rows leaving the window during a stage are exactly the rows that ENTERED it one
// stage earlier -- the same lane's rows of the previous stage, still in its registers.  The second read stream (7.2 GB at C4,
// re-fetched from HBM because the output stream had evicted it: profiles/r02_traffic.json) and its half of the LDS image and
// of the LDS -> register hand-over disappear; only the first stage of a tile fetches its leaving rows (straight into registers).
template <typename T, int PP, int MODE, int FULLP, int OLDREG = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void rolling_seg_kernel(
    const T* const* __restrict__ cols, RollArgs ra_in, const double* __restrict__ tile_tot, T* __restrict__ coeffs,
    T* __restrict__ pred, uint8_t* __restrict__ valid) {
    static_assert(!OLDREG || MODE == 0, "the register-resident leaving rows belong to the rolling form");
    using SD = SegDims<T, PP, OLDREG>;
    constexpr int kTile = MODE == 0 ? kSegTileRoll : kSegTile;
    constexpr int K = kSegK, NG = SD::NG, NV = SD::NV, E16 = SD::E16, PIECES = SD::PIECES;
    static_assert(NV <= 64, "one lane per moment in the scan");
    static_assert(MODE == 0 || MODE == 2, "rolling or the main pass of the expanding fit");
    using V16 = typename Tile<T>::vec;  // 16 bytes, element aligned (d2u / f4u)
    RollArgs ra = ra_in;
    if constexpr (FULLP == 1) {
        ra.p = PP;
        ra.pp = PP;
        ra.bias = 0;
    } else if constexpr (FULLP == 2) {
        ra.p = PP - 1;
        ra.pp = PP;
        ra.bias = 1;
    }
    const int p = ra.p;                       // feature columns; column p of the table is y
    constexpr int NWHICH = (MODE == 0 && !OLDREG) ? 2 : 1; // new rows (+ the rows leaving the window, unless they are register resident)
    constexpr int NLOAD = NWHICH * (PP + 1) * PIECES;       // 16-byte loads per lane and stage (columns beyond p are skipped)
    constexpr int PER_BATCH = (NLOAD + K - 1) / K;          // ... issued in K batches, one in front of every row of pass 2
    extern __shared__ __attribute__((aligned(16))) double seg_lds[];  // (one name / type per translation unit)
    // explicit LDS address space: with the profiling statements in the way the compiler no longer inferred it for every access and
    // emitted generic stores (whose aperture check it then failed to select: "Illegal instruction ... src_shared_base")
    typedef __attribute__((address_space(3))) char* lds_c;
    typedef __attribute__((address_space(3))) double* lds_dp;
    lds_c sm = (lds_c)reinterpret_cast<char*>(seg_lds);
    lds_dp D = (lds_dp)seg_lds;
    const int lane = threadIdx.x & 63;
    const int64_t n = ra.n, w = ra.window;
    const int64_t ntiles = (n + kTile - 1) / kTile;
    const double lambda = ra.lambda;

    // ---- one 16-byte piece of the next stage: global -> register (issue) -> LDS (commit)
    auto piece_row = [&](int idx, int64_t base) __attribute__((always_inline)) {
        const int piece = idx % PIECES, which = idx / (PIECES * (PP + 1));
        return base + (int64_t)piece * (64 * E16) + (int64_t)lane * E16 - (which ? w : 0);
    };
    auto issue = [&](int idx, int64_t base, V16& v) __attribute__((always_inline)) {
        const int c = (idx / PIECES) % (PP + 1);
        if (c > p) return;  // (unused column slot)
        const int64_t r = piece_row(idx, base);
        gptr<T> col = as_global(cols[c]);
        if (r >= 0 && r + E16 <= n) {
            v = *reinterpret_cast<gptr<V16>>(col + r);
        } else {
#pragma unroll
            for (int e = 0; e < E16; ++e) v[e] = (r + e >= 0 && r + e < n) ? col[r + e] : T(0);
        }
    };
    auto commit = [&](int idx, const V16& v) __attribute__((always_inline)) {
        const int c = (idx / PIECES) % (PP + 1), piece = idx % PIECES, which = idx / (PIECES * (PP + 1));
        if (c > p) return;
        *(__attribute__((address_space(3))) V16*)(sm + (which * (PP + 1) + c) * SD::STREAM_BYTES + piece * 1024 + lane * 16) = v;
    };
#ifdef PDS_ROLL_LDS_DIRECT
    // gfx950: global_load_lds_dwordx4 -- the 16 bytes of lane L land at (wave-uniform LDS base) + 16 L without passing
    // through a register: the next stage's image is fetched by ONE burst of asynchronous loads at the start of pass 2 and is
    // complete long before the next stage reads it (no staging registers, no ds_write commits, no wait behind every row)
    auto issue_direct = [&](int idx, int64_t base) __attribute__((always_inline)) {
        const int c = (idx / PIECES) % (PP + 1), piece = idx % PIECES, which = idx / (PIECES * (PP + 1));
        if (c > p) return;
        const int64_t r = piece_row(idx, base);
        gptr<T> col = as_global(cols[c]);
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* glb_ptr;
        __builtin_amdgcn_global_load_lds((glb_ptr)(col + r), (lds_ptr)(sm + (which * (PP + 1) + c) * SD::STREAM_BYTES + piece * 1024),
                                         16, 0, 0);
    };
#endif
    // the lane's K rows of stream (which, c) out of the LDS image
    auto pick = [&](int which, int c, double (&out)[K]) __attribute__((always_inline)) {
        const lds_c src = sm + (which * (PP + 1) + c) * SD::STREAM_BYTES + lane * (K * (int)sizeof(T));
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const V16 v = *(const __attribute__((address_space(3))) V16*)(src + 16 * j);
#pragma unroll
            for (int e = 0; e < E16; ++e) out[j * E16 + e] = (double)v[e];
        }
    };

#ifdef PDS_PROFILE_ROLLING
    unsigned long long rprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    // the lane's rows of the stage (new) and the rows leaving the window with them (old); OLDREG: `rn` / `okn` of one stage
    // become `ro` / `oko` of the next, so they live across the stage loop
    SegRow<PP> rn[K], ro[K];
    bool okn[K], oko[K];
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        RT0();
        const int64_t t0 = t * kTile, t1 = (t0 + kTile < n) ? t0 + kTile : n;
        // ---- anchor: lane v < NV carries moment v of the window that ends at row t0 - 1
        double carry = 0.0;
        if constexpr (MODE == 2) {
            if (lane < NV) carry = tile_tot[t * NV + lane];
        } else {
            double A[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) A[v] = 0.0;
            const int64_t a0 = t0 - w;
            for (int64_t r = a0 + lane; r < t0; r += 64) {
                SegRow<PP> row;
                const bool in = r >= 0;
                // unconditional loads from clamped (column, row): a load under its own exec mask is serialised behind the previous
                // one by the compiler's vmcnt(0) (keyed_partition.hip found the same pattern)
                const int64_t rc = in ? r : 0;
#pragma unroll
                for (int c = 0; c < PP; ++c) {
                    const double x = (double)as_global(cols[c < p ? c : p])[rc];
                    row.z[c] = (c < p) ? (in ? x : 0.0) : ((c == p && ra.bias) ? 1.0 : 0.0);
                }
                row.y = in ? (double)as_global(cols[p])[rc] : 0.0;
                const bool ok = in && seg_finite<PP>(row);
                if (!ok) seg_zero<PP>(row);
                seg_accumulate<PP, NV, 1>(A, row, ok);
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) D[v * kSegStride + lane] = A[v];
            PDS_WAVE_LDS_SYNC();
            if (lane < NV) {
                const lds_dp rowp = D + lane * kSegStride;
                double s = 0.0;
#pragma unroll 16
                for (int i = 0; i < 64; ++i) s += rowp[i];
                carry = s;
            }
            PDS_WAVE_LDS_SYNC();
        }
        // ---- first stage of the tile: straight through (issue everything, commit everything)
        {
            V16 tmp[NLOAD];
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) issue(i, t0, tmp[i]);
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) commit(i, tmp[i]);
        }
        if constexpr (OLDREG) {
            // the rows in front of the tile ([t0 - w, t0), w == kSegStage: this lane's K rows of the stage before the first)
            // play the previous stage: straight from global memory, guarded at the front of the frame
            const int64_t rp = t0 - kSegStage + (int64_t)K * lane;
            auto load_prev = [&](int cc, double (&v)[K]) __attribute__((always_inline)) {
                // branch-free (tiles start at multiples of the tile length, so the K rows are all inside the frame or all in front of
                // it): the loads of the PP + 1 columns go out back to back instead of one round trip per column
                gptr<T> col = as_global(cols[cc]);
                const int64_t rq = rp >= 0 ? rp : 0;
#pragma unroll
                for (int j = 0; j < PIECES; ++j) {
                    const V16 x = *reinterpret_cast<gptr<V16>>(col + rq + j * E16);
#pragma unroll
                    for (int e = 0; e < E16; ++e) v[j * E16 + e] = rp >= 0 ? (double)x[e] : 0.0;
                }
            };
#pragma unroll
            for (int c = 0; c < PP; ++c) {
                double v[K];
                load_prev(c < p ? c : p, v);  // (unconditional: see the anchor loop)
#pragma unroll
                for (int i = 0; i < K; ++i) rn[i].z[c] = (c < p) ? v[i] : ((c == p && ra.bias) ? 1.0 : 0.0);
            }
            {
                double v[K];
                load_prev(p, v);
#pragma unroll
                for (int i = 0; i < K; ++i) rn[i].y = v[i];
            }
#pragma unroll
            for (int i = 0; i < K; ++i) okn[i] = (rp + i >= 0) && seg_finite<PP>(rn[i]);
        }
        RT1(0);  // anchor + first stage straight through
        for (int64_t base = t0; base < t1; base += kSegStage) {
            RTA();
            PDS_WAVE_LDS_SYNC();  // the stage image is complete
            const int64_t r0 = base + (int64_t)K * lane;
            // ---- the lane's rows: LDS -> registers
            if constexpr (OLDREG) {
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    ro[i] = rn[i];
                    oko[i] = okn[i] && (r0 + i < t1);
                }
            }
#pragma unroll
            for (int c = 0; c < PP; ++c) {
                double a[K], b[K];
                const bool feat = c < p;
                if (feat) {
                    pick(0, c, a);
                    if constexpr (MODE == 0 && !OLDREG) pick(1, c, b);
                }
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const double one = (c == p && ra.bias) ? 1.0 : 0.0;
                    rn[i].z[c] = feat ? a[i] : one;
                    if constexpr (!OLDREG) ro[i].z[c] = (MODE == 0) ? (feat ? b[i] : one) : 0.0;
                }
            }
            {
                double a[K], b[K];
                pick(0, p, a);
                if constexpr (MODE == 0 && !OLDREG) pick(1, p, b);
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    rn[i].y = a[i];
                    if constexpr (!OLDREG) ro[i].y = (MODE == 0) ? b[i] : 0.0;
                }
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int64_t r = r0 + i;
                // (a row that does not count -- outside the tile, in front of the frame, holding a non-finite value -- is
                //  skipped by an exec-mask branch around its accumulation: 18 selects per row saved over zeroing its values)
                okn[i] = (r < t1) && seg_finite<PP>(rn[i]);
                if constexpr (MODE == 0 && !OLDREG) {
                    oko[i] = (r < t1) && (r - w >= 0) && seg_finite<PP>(ro[i]);
                } else if constexpr (MODE != 0) {
                    oko[i] = false;
                }
            }
#ifdef PDS_PROFILE_ROLLING
            asm volatile("" :: "v"(rn[0].z[0]), "v"(rn[K - 1].y));
#endif
            RT1(1);  // rows LDS -> registers, finiteness
            RTA();
            // ---- pass 1: the lane's own increments
            double S[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) S[v] = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (okn[i]) seg_accumulate<PP, NV, 1>(S, rn[i], true);
                if constexpr (MODE == 0) {
                    if (oko[i]) seg_accumulate<PP, NV, -1>(S, ro[i], true);
                }
            }
#ifdef PDS_PROFILE_ROLLING
            asm volatile("" :: "v"(S[0]), "v"(S[NV - 1]));
#endif
            RT1(2);  // pass 1
            RTA();
            // ---- scan over the lanes through LDS (every lane holds its rows in registers: the region is free)
            PDS_WAVE_LDS_SYNC();
#pragma unroll
            for (int v = 0; v < NV; ++v) D[v * kSegStride + lane] = S[v];
            PDS_WAVE_LDS_SYNC();
            if (lane < NV) {
                lds_dp rowp = D + lane * kSegStride;
                double run = carry;
#pragma unroll
                for (int i0 = 0; i0 < 64; i0 += 16) {
                    double x[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) x[i] = rowp[i0 + i];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const double inc = x[i];
                        x[i] = run;  // exclusive prefix, carry included
                        run += inc;
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) rowp[i0 + i] = x[i];
                }
                carry = run;
            }
            PDS_WAVE_LDS_SYNC();
#pragma unroll
            for (int v = 0; v < NV; ++v) S[v] = D[v * kSegStride + lane];
            PDS_WAVE_LDS_SYNC();  // the region is free again: the next stage's pieces may land
            RT1(3);  // scan
            RTA();
            // ---- pass 2: row by row; a quarter of the next stage's pieces goes in flight in front of every row
            const int64_t nb = base + kSegStage;
#ifdef PDS_ROLL_LDS_DIRECT
            // interior stages (every piece of both streams inside the frame): the asynchronous burst; frame edges keep the
            // guarded register path below
            const bool direct = (nb < t1) && (nb + kSegStage <= n) && (MODE != 0 || OLDREG || nb - w >= 0);
            const bool edge = (nb < t1) && !direct;  // (fetched behind the rows, in small batches: rare and allowed to wait)
            const bool more = false;
            if (direct) {
#pragma unroll
                for (int idx = 0; idx < NLOAD; ++idx) issue_direct(idx, nb);
            }
#else
            const bool more = nb < t1;
#endif
            auto diag_add = [&](int a) __attribute__((always_inline)) { return (a < ra.pp) ? lambda : 1.0; };
#define GI(g, a, b) g[(a) * PP - ((a) * ((a)-1)) / 2 + ((b) - (a))]
            // step 0 of the L D L' of (G + lambda I) reads the running sums and writes the work copy (no register copy of the NG
            // values); idx(a, b), a <= b -> a * PP - a (a - 1) / 2 + (b - a).  Padding dimensions (a >= p') have zero rows
            // and a unit diagonal: beta_pad = 0.
            auto fact_first = [&](double (&g)[NG], double (&c)[PP], double (&rd)[PP], bool& okc) __attribute__((always_inline)) {
#pragma unroll
                for (int a = 0; a < PP; ++a) c[a] = S[NG + a];
                const double d = GI(S, 0, 0) + diag_add(0);
                okc = d > 0.0;
                double x = __builtin_amdgcn_rcp(d);
#pragma unroll
                for (int it = 0; it < PDS_RCP_NEWTON; ++it) x = x * fma(-d, x, 2.0);
                rd[0] = x;
#pragma unroll
                for (int a = 1; a < PP; ++a) {
                    const double tka = GI(S, 0, a) * x;
#pragma unroll
                    for (int b = a; b < PP; ++b) {
                        // (no `+ 0.0` on the off-diagonal entries: the compiler must keep such an add -- it turns -0.0 into
                        //  +0.0 -- and it cost 28 f64 instructions per row)
                        if (a == b) GI(g, a, b) = fma(-tka, GI(S, 0, b), GI(S, a, b) + diag_add(a));
                        else GI(g, a, b) = fma(-tka, GI(S, 0, b), GI(S, a, b));
                    }
                    GI(g, 0, a) = tka;
                }
            };
            auto fact_step = [&](int k, double (&g)[NG], double (&rd)[PP], bool& okc) __attribute__((always_inline)) {
                const double d = GI(g, k, k);
                okc = okc && (d > 0.0);
                double x = __builtin_amdgcn_rcp(d);
#pragma unroll
                for (int it = 0; it < PDS_RCP_NEWTON; ++it) x = x * fma(-d, x, 2.0);
                rd[k] = x;
#pragma unroll
                for (int a = k + 1; a < PP; ++a) {
                    const double tka = GI(g, k, a) * x;  // l_ak
#pragma unroll
                    for (int b = a; b < PP; ++b) GI(g, a, b) = fma(-tka, GI(g, k, b), GI(g, a, b));
                    GI(g, k, a) = tka;
                }
            };
            // L u = c, D v = u, L' beta = v -- one substitution step at a time, so that two systems can be walked in lockstep
            auto fwd_step = [&](int a, const double (&g)[NG], double (&c)[PP]) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < PP; ++k)
                    if (k < a) c[a] = fma(-GI(g, k, a), c[k], c[a]);
            };
            auto bwd_step = [&](int a, const double (&g)[NG], double (&c)[PP]) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < PP; ++k)
                    if (k > a) c[a] = fma(-GI(g, a, k), c[k], c[a]);
            };
            auto emit = [&](int i, const double (&c)[PP], bool okc, double cnt) __attribute__((always_inline)) {
                const int64_t r = r0 + i;
                if (r < t1) {
                    const T nanv = (T)__builtin_nan("");
                    bool v_ok = r >= w - 1;
                    if (ra.min_size > 0) v_ok = v_ok && (cnt >= (double)ra.min_size);
                    double pr = 0.0;
#pragma unroll
                    for (int a = 0; a < PP; ++a) pr = fma(rn[i].z[a], c[a], pr);
                    const bool good = v_ok && okc;
                    T* out = coeffs + r * (int64_t)ra.pp;
                    if constexpr (FULLP != 0 && PP % E16 == 0) {
                        // p' == PP: the row is PP contiguous values -> 16-byte stores (element-aligned vector type)
#pragma unroll
                        for (int a = 0; a < PP; a += E16) {
                            V16 o;
#pragma unroll
                            for (int e = 0; e < E16; ++e) o[e] = good ? (T)c[a + e] : nanv;
                            *reinterpret_cast<V16*>(out + a) = o;
                        }
                    } else {
#pragma unroll
                        for (int a = 0; a < PP; ++a)
                            if (a < ra.pp) out[a] = good ? (T)c[a] : nanv;
                    }
                    pred[r] = (good && okn[i]) ? (T)pr : nanv;  // (a non-finite row: x_r . beta is NaN in the reference too)
                    valid[r] = v_ok ? 1 : 0;
                }
            };
            auto advance = [&](int i) __attribute__((always_inline)) {  // the window moves on by row i of the lane
                if (okn[i]) seg_accumulate<PP, NV, 1>(S, rn[i], true);
                if constexpr (MODE == 0) {
                    if (oko[i]) seg_accumulate<PP, NV, -1>(S, ro[i], true);
                }
            };
#ifdef PDS_ROLL_PAIR
            // Two rows' factorisations in lockstep: the dependent chains of one (pivot reciprocal + two Newton steps, the
            // substitution recurrences) fill the latency slots of the other.  Measured: 4.79 vs 4.80 ms at C4 -- the scheduler
            // already overlaps what can be overlapped; kept behind the flag for the record (profiles/r02_rolling_variants_ab.txt).
            static_assert(K % 2 == 0, "rows are walked in pairs");
#pragma unroll
            for (int i = 0; i < K; i += 2) {
#if defined(PDS_ROLL_LDS_DIRECT) && defined(PDS_ROLL_WAIT_LAST_ROW)
                if (i == K - 2 && direct) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (see the single-row form below)
#endif
                double g0[NG], g1[NG], c0[PP], c1[PP], rd0[PP], rd1[PP];
                bool ok0 = true, ok1 = true;
                advance(i);
                const double cnt0 = S[NV - 1];
                fact_first(g0, c0, rd0, ok0);
                advance(i + 1);
                const double cnt1 = S[NV - 1];
                fact_first(g1, c1, rd1, ok1);
#pragma unroll
                for (int k = 1; k < PP; ++k) {
                    fact_step(k, g0, rd0, ok0);
                    fact_step(k, g1, rd1, ok1);
                }
#pragma unroll
                for (int a = 1; a < PP; ++a) {
                    fwd_step(a, g0, c0);
                    fwd_step(a, g1, c1);
                }
#pragma unroll
                for (int a = 0; a < PP; ++a) {
                    c0[a] *= rd0[a];
                    c1[a] *= rd1[a];
                }
#pragma unroll
                for (int a = PP - 2; a >= 0; --a) {
                    bwd_step(a, g0, c0);
                    bwd_step(a, g1, c1);
                }
#if defined(PDS_ROLL_LDS_DIRECT) && !defined(PDS_ROLL_WAIT_LAST_ROW)
                if (i == 0 && direct) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (see the single-row form below)
#endif
                emit(i, c0, ok0, cnt0);
                emit(i + 1, c1, ok1, cnt1);
            }
            (void)more;
#else
#pragma unroll
            for (int i = 0; i < K; ++i) {
                V16 tmp[PER_BATCH];
                if (more) {
#pragma unroll
                    for (int j = 0; j < PER_BATCH; ++j)
                        if (i * PER_BATCH + j < NLOAD) issue(i * PER_BATCH + j, nb, tmp[j]);
                }
#if defined(PDS_ROLL_LDS_DIRECT) && defined(PDS_ROLL_WAIT_LAST_ROW)
                // (round 2's placement, kept for the A/B: in front of the LAST row -- but vmcnt counts stores as well, so this
                //  also waited for the three rows of coefficient stores issued since the burst to reach memory)
                if (i == K - 1 && direct) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                advance(i);
                double g[NG], c[PP], rd[PP];
                bool okc = true;
                fact_first(g, c, rd, okc);
#pragma unroll
                for (int k = 1; k < PP; ++k) fact_step(k, g, rd, okc);
#pragma unroll
                for (int a = 1; a < PP; ++a) fwd_step(a, g, c);
#pragma unroll
                for (int a = 0; a < PP; ++a) c[a] *= rd[a];
#pragma unroll
                for (int a = PP - 2; a >= 0; --a) bwd_step(a, g, c);
#if defined(PDS_ROLL_LDS_DIRECT) && !defined(PDS_ROLL_WAIT_LAST_ROW)
                // The compiler does not order LDS reads behind global_load_lds (its ISA for this kernel has no vmcnt wait in
                // front of the next stage's ds_reads), so the wave waits itself.  vmcnt counts loads AND stores (in issue order),
                // so the wait sits where no store of this stage has been issued yet: behind the first row's arithmetic (the burst
                // has had a row's time, ~3 500 clk), in front of its stores -- the stores of all four rows then drain behind the
                // following rows' arithmetic instead of being waited for.
                if (i == 0 && direct) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                emit(i, c, okc, S[NV - 1]);
                if (more) {
#ifdef PDS_PROFILE_ROLLING
                    const unsigned long long _tc = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
                    for (int j = 0; j < PER_BATCH; ++j)
                        if (i * PER_BATCH + j < NLOAD) commit(i * PER_BATCH + j, tmp[j]);
#ifdef PDS_PROFILE_ROLLING
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    rprof[5] += __builtin_amdgcn_s_memtime() - _tc;
#endif
                }
            }
#endif
#undef GI
#ifdef PDS_ROLL_LDS_DIRECT
            if (edge) {
#pragma unroll 1
                for (int b = 0; b < K; ++b) {
                    V16 tmpe[PER_BATCH];
#pragma unroll
                    for (int j = 0; j < PER_BATCH; ++j)
                        if (b * PER_BATCH + j < NLOAD) issue(b * PER_BATCH + j, nb, tmpe[j]);
#pragma unroll
                    for (int j = 0; j < PER_BATCH; ++j)
                        if (b * PER_BATCH + j < NLOAD) commit(b * PER_BATCH + j, tmpe[j]);
                }
            }
#endif
            RT1(4);  // pass 2 (incl. the commits counted in [5])
        }
        PDS_WAVE_LDS_SYNC();  // the next tile's anchor writes the region
    }
#ifdef PDS_PROFILE_ROLLING
    rprof[7] = __builtin_amdgcn_s_memtime() - t_begin;
    if (lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_roll_cycles[k], rprof[k]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Expanding fit, pass 1: the moment totals of every 4096-row tile -- a pure streaming reduction (45 FMAs per row at p' = 8
// against 72 bytes: HBM bound), so it reads like the Gram kernels: 16 bytes per lane straight down each column (1 KiB
// coalesced per instruction, non-temporal), moments in registers, three waves per SIMD,
// one cross-lane reduction per tile.  (It was the lane = row rolling kernel in its totals mode: 3.06 ms for the 7.2 GB of
// C4's frame.)  Same moment order and non-finite rule as rolling_seg_kernel (seg_accumulate).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int PP, int FULLP>
__global__ __launch_bounds__(64) void rolling_totals_kernel(const T* const* __restrict__ cols, RollArgs ra_in,
                                                            double* __restrict__ tile_tot) {
    using SD = SegDims<T, PP>;
    constexpr int NV = SD::NV, E16 = SD::E16;
    constexpr int STEP = 64 * E16;  // rows per load step
    using V16 = typename Tile<T>::vec;
    RollArgs ra = ra_in;
    if constexpr (FULLP == 1) {
        ra.p = PP;
        ra.pp = PP;
        ra.bias = 0;
    } else if constexpr (FULLP == 2) {
        ra.p = PP - 1;
        ra.pp = PP;
        ra.bias = 1;
    }
    const int p = ra.p;
    extern __shared__ __attribute__((aligned(16))) double tot_lds[];
    const int lane = threadIdx.x & 63;
    const int64_t n = ra.n;
    const int64_t ntiles = (n + kSegTile - 1) / kSegTile;
    auto load_step = [&](int64_t row, V16 (&v)[PP + 1]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c <= PP; ++c) {
            if (c > p) continue;
            gptr<T> col = as_global(cols[c]);
            if (row + E16 <= n) {
                v[c] = __builtin_nontemporal_load(reinterpret_cast<gptr<V16>>(col + row));
            } else {
#pragma unroll
                for (int e = 0; e < E16; ++e) v[c][e] = (row + e < n) ? col[row + e] : T(0);
            }
        }
    };
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t t0 = t * kSegTile, t1 = (t0 + kSegTile < n) ? t0 + kSegTile : n;
        double S[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) S[v] = 0.0;
        // One register set, three waves per SIMD (142 VGPRs), the next step's loads issued after this step's math: A/B against
        // a register double buffer at two waves per SIMD (212 VGPRs) 1.12 vs 1.36 ms for C4's 7.2 GB = 0.80 vs 0.66 of the HBM
        // peak -- more resident waves hide the latency better than a deeper burst per wave.
        V16 cur[PP + 1];
        for (int64_t base = t0; base < t1; base += STEP) {
            load_step(base + (int64_t)lane * E16, cur);
#pragma unroll
            for (int e = 0; e < E16; ++e) {
                SegRow<PP> row;
#pragma unroll
                for (int c = 0; c < PP; ++c) row.z[c] = (c < p) ? (double)cur[c][e] : ((c == p && ra.bias) ? 1.0 : 0.0);
                row.y = (double)cur[p][e];
                const bool ok = (base + (int64_t)lane * E16 + e < t1) && seg_finite<PP>(row);
                if (!ok) seg_zero<PP>(row);
                seg_accumulate<PP, NV, 1>(S, row, ok);
            }
        }
        // ---- the tile's totals: sum over the lanes in lane order (fixed order: reproducible)
#pragma unroll
        for (int v = 0; v < NV; ++v) tot_lds[v * kSegStride + lane] = S[v];
        PDS_WAVE_LDS_SYNC();
        if (lane < NV) {
            const double* rowp = tot_lds + lane * kSegStride;
            double s = 0.0;
#pragma unroll 16
            for (int i = 0; i < 64; ++i) s += rowp[i];
            tile_tot[t * NV + lane] = s;
        }
        PDS_WAVE_LDS_SYNC();
    }
}

}  // namespace pds
