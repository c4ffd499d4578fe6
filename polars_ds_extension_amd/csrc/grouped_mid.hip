// grouped_mid.hip -- grouped regressions with 17 .. 64 f64 features (17 .. 32 also f32) as ONE stream over the frame: the (p+2)^2 moment
// records of contiguous groups (33 .. 64 features, and the fallback), or -- 17 .. 32 features -- no records at all: a streaming and a
// solving wave per SIMD, the finished group crossing an LDS slot.  The half-tile layout is moments_mid.hip's
// (moments_mid_dev.hpp).
#include "common.hpp"
#include "moments_dev.hpp"
#include "moments_mid_dev.hpp"
#include "solve_wave_dev.hpp"
#include "solve_row16_dev.hpp"

#ifndef PDS_MID_DIRECT
#define PDS_MID_DIRECT 1
#endif
// the direct form's loads: two consecutive instructions read the two halves of the same sixteen 128-byte lines -- as non-temporal loads
// (-DPDS_MID_DIRECT_NT) the second one misses again: 30 features 6.4 -> 7.5 ms (profiles/r06_grouped_mid_direct.txt)
// (A/B: 1 = the lane terms of the slot / record addresses pass through an empty asm in every direct kernel, not only where registers are short)
#ifndef PDS_MID_LAUNDER_ALL
#define PDS_MID_LAUNDER_ALL 0
#endif
// the record stream of 33 .. 64 features in the direct form (0: the LDS form of rounds 3 - 5)
#ifndef PDS_MID_DIRECT_RECORDS
#define PDS_MID_DIRECT_RECORDS 1
#endif
#ifndef PDS_MID_DIRECT_OCTET
#define PDS_MID_DIRECT_OCTET 1
#endif
// a group that crosses a wave boundary is finished by the wave it starts in (0: summed by both waves into a side record, rounds 4 / 5)
#ifndef PDS_MID_OWNER
#define PDS_MID_OWNER 1
#endif
#ifdef PDS_MID_DIRECT_NT
#define PDS_MID_DIRECT_LOAD(q) __builtin_nontemporal_load(q)
#else
#define PDS_MID_DIRECT_LOAD(q) (*(q))
#endif

namespace pds {

// -DPDS_PROFILE_MID: per-phase shader-clock sums of the paired grouped stream's waves (development; tools/grouped_mid_profile.py)
#ifdef PDS_PROFILE_MID
__device__ unsigned long long g_mid_phase[16];
#define PDS_MT(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PDS_MADD(k, t0) mprof[k] += __builtin_amdgcn_s_memtime() - (t0)
#else
#define PDS_MT(var) do {} while (0)
#define PDS_MADD(k, t0) do {} while (0)
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------------------------------
// Grouped form: the (p+2)^2 moment records of CONTIGUOUS GROUPS (group g = rows [off[g], off[g+1])) from ONE stream over the frame --
// `group_by(key).agg(pds.lin_reg(...))` with 17 .. 64 f64 features.  The one-wave-per-group kernel of moments.hip loads 32 rows at a
// time with 8-byte loads and nothing in flight behind them (1.2 - 2.5 TB/s); here the rows are read exactly as above (waves own
// contiguous, half-tile aligned row ranges; 1 KiB asynchronous loads into wave-private LDS images) and the accumulators are CUT at
// group boundaries: a half-tile is walked as segments [lo, hi) of one group each, the first and last 4-row step of a segment with the
// rows outside it zeroed in the operands, and a finished group's tiles go straight from the accumulator registers into its record.
// Groups that lie inside one wave's rows are written with plain stores; a group cut by a wave boundary (two per wave, or a giant group
// over many waves) is added to the zero-initialised record with atomics.
// SPPC > 0 (NBLK = 2, up to 32 features, round 4): NO RECORDS -- a finished group is solved in the wave that streamed it.  Its
// accumulator tiles go through a 4 KB LDS scratch (16 columns per trip) into one of four pending systems -- one per 16-lane DPP row,
// lane t = columns t and 16 + t, centred -- and four pending systems are factored side by side (solve_row16_dev.hpp: L D L' with the
// pivot-ratio gate) and their coefficients written; only what cannot be answered in place leaves as a record: a group cut by a wave
// boundary (atomics into the side table's slot of the wave it starts in: at most one per wave) and a system next to the gate
// (appended to the marked list for the pivoted QR; its record is rebuilt from its rows).  The per-group dispatch
// of pl_lr under group_by (linear_regression.rs:447-497) at 17 .. 32 features then moves input + coefficients only, as at <= 16.
template <typename T>
struct MidSolveArgsT {
    SolveRegDev sp;
    T* coeffs = nullptr;           // [n_groups][p + bias]
    uint8_t* flags = nullptr;      // [n_groups]: 1 = null
    double* side_rec = nullptr;    // [waves][q * q], zeroed: records of groups that straddle wave boundaries
    int32_t* side_list = nullptr;  // [waves], -1: slot w = the group that starts in wave w's rows and ends beyond them
    double* mark_rec = nullptr;    // [mark_cap][q * q]: records of systems the in-wave solve marked
    int32_t* mark_list = nullptr;  // [mark_cap]
    unsigned* mark_count = nullptr;  // appended count (beyond mark_cap: overflow, the caller falls back to the record pipeline)
    unsigned mark_cap = 0;
};
using MidSolveArgs = MidSolveArgsT<double>;
constexpr int kMidSolveScratch = 5120;  // bytes behind the tile images (4 x 40 KB per CU): 24 columns x 26 doubles in one trip, or 16 x 34 per trip
// v summed lane-wise over the four 16-lane rows of the wave (every lane gets its column's total): v_permlane16_swap / v_permlane32_swap
// of gfx950 -- with both operands the same value the swap leaves [r0 r0 r2 r2] and [r1 r1 r3 r3] (rows), then the two halves -- four
// vector moves per stage instead of two trips through the LDS crossbar (__shfl_xor)
__device__ __forceinline__ double rows_sum4(double v) {
    {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    return v;
}
// the slot between a streaming and a solving wave (PAIRED): the upper triangle of the SPPC x SPPC moment block row by row, U(r, c) at
// tri(r) + c - r, then X'y, the column sums, and [rows, sum y, group] -- 4 760 bytes at 32 features
// YC: the triangle is that of [X 1 y]' [X 1 y] (QC = columns incl. the two), only the group id follows it.
template <int SPPC, bool YC = false>
struct MidPacked {
    static constexpr int QC = YC ? (SPPC + 2 < 32 ? SPPC + 2 : 32) : SPPC;
    static constexpr int tri(int r) { return r * QC - r * (r - 1) / 2; }
    static __device__ __forceinline__ int tri_rt(int r) { return r * QC - ((r * (r - 1)) >> 1); }
    static constexpr int XY = QC * (QC + 1) / 2, CS = XY + (YC ? 0 : SPPC), TAIL = CS + (YC ? 0 : SPPC), GID = TAIL + (YC ? 0 : 2), COUNT = GID + 1;
};
// DIRECT: the LDS the tile images no longer take holds a RING of slots between the streaming and the solving wave -- a finished group waits
// there, not in a spare register set of the streaming wave (which has none left beside its two operand sets)
constexpr int kMidDirectSlots = 8;       // two batches of four: the solving wave takes FOUR groups at once (one per 16-lane row, one read per value)
constexpr int kMidDirectSlotBytes = 4768;  // the upper triangle of 32 columns, X'y, the column sums, [rows, sum y], the group id: to 16 bytes
template <int ES>
constexpr int kMidDirectRows = ES == 8 ? 64 : 128;  // rows per half-tile of the direct form: eight blocks of 16 bytes per lane and slot
constexpr int kMidDirectYBytes = 64 * 8 + 16;  // one image of the target's half-tile (31 / 32 features: the target does not fit the operand blocks)
template <int NBLK, bool DIRECT = false>
constexpr int kMidPairLds = DIRECT ? kMidDirectSlots * kMidDirectSlotBytes + 2 * kMidDirectYBytes + 64  // the direct form: a ring of slots, two target images, flag words
                                      : MidDims<NBLK>::LDS_BYTES + kMidSolveScratch + 64;  // tile images + scratch + flag words of one pair of waves (PAIRED)
// DIRECT: the ones column and the padding columns of the second operand block are loaded like every other column -- from 64 ones and
// 64 zeros, with a lane stride of nothing per half-tile (no selects, no partially active load instructions)
#define PDS_R8(v) v, v, v, v, v, v, v, v
__device__ const double g_mid_direct_const[128] = {PDS_R8(1.0), PDS_R8(1.0), PDS_R8(1.0), PDS_R8(1.0), PDS_R8(1.0), PDS_R8(1.0), PDS_R8(1.0), PDS_R8(1.0)};
__device__ const float g_mid_direct_const_f32[256] = {PDS_R8(1.0f), PDS_R8(1.0f), PDS_R8(1.0f), PDS_R8(1.0f), PDS_R8(1.0f), PDS_R8(1.0f), PDS_R8(1.0f), PDS_R8(1.0f),
                                                      PDS_R8(1.0f), PDS_R8(1.0f), PDS_R8(1.0f), PDS_R8(1.0f), PDS_R8(1.0f), PDS_R8(1.0f), PDS_R8(1.0f), PDS_R8(1.0f)};
#undef PDS_R8

// PAIRED (with SPPC): a workgroup is FOUR PAIRS of waves -- waves 0 .. 3 stream (loads, matrix steps, group walk: what a wave of the
// unpaired form does up to the finished group), waves 4 .. 7 are their solvers: a finished group's moments cross the pair's LDS scratch
// (a sequence-numbered slot: full / taken / done words behind it, polled with s_sleep -- no workgroup barrier after the first), the
// solver wave keeps the four pending systems, factors them and writes coefficients, flags and marks.  Waves w and w + 4 of a workgroup
// share a SIMD (tools/wave_placement.hip), so every SIMD runs one streaming and one solving wave: the hand-over and the solves of one
// overlap the load waits and matrix instructions of the other (they ran one after the other in a single wave: 2.5 of 7.65 ms).
// YC (PAIRED, up to 30 features): the ones and the target are columns p and p + 1 of the second operand block -- the column sums, X'y, the
// row count and sum y come out of the matrix instructions that run anyway (no side sums per step, no cross-row reductions per group),
// and the accumulator blocks ARE the record [X 1 y]' [X 1 y].
// NQ = 1 (YC, up to 18 features: the second operand block holds at most FOUR columns -- x16, x17, the ones, the target): its two
// 16 x 16 x 4 matrix instructions per step (64 ticks each, a quarter of a block useful) become two v_mfma_f64_4x4x4_4b (20 ticks each; layout
// and rate: tools/mfma_f64_4x4_probe.hip -- A[i][k] of block b in lane 16 k + 4 b + i, B[k][j] in lane 16 k + 4 b + j, D[i][j] in lane
// 16 i + 4 b + j): block b of the first multiplies columns 4 b .. 4 b + 3 of the first operand block (its lanes ARE the 16 x 16 operand's)
// with the quad, the second the quad with itself.
// T = float (PAIRED only): f32 frames -- 128-row half-tiles of the same 1 KiB instructions and the same LDS bytes, widened to f64 on their
// way out of LDS; moments, slot, solve and side / marked records are f64 as for f64 frames, the coefficients are written as T.
// DIRECT (PAIRED; f64 frames of 17 .. 32 features, f32 frames of 17 .. 30; round 6): NO tile images.  The streaming wave loads the half-tile straight into the matrix instructions'
// operand layout -- lane (feature = lane % 16, slot = lane / 16) reads 16 bytes = rows 8 k + 2 slot, + 1 of its own column for block k of
// eight rows: sixteen columns x 64 contiguous bytes per 1 KiB load instruction, the order of the rows inside a block does not matter to a
// sum over rows -- into one of two register sets (the half-tile being walked, the next one in flight); a 4-row step multiplies rows
// {8 k + 2 slot + j}.  No asynchronous LDS pieces (95 clk of the wave's time each), no operand reads from LDS in front of every step.
template <int NBLK, int SPPC = 0, bool PAIRED = false, bool YC = false, int NQ = 0, typename T = double, bool DIRECT = false>
__global__ __launch_bounds__(PAIRED ? 512 : 64) void grouped_mid_stream_kernel(const T* const* __restrict__ cols, int p, int64_t n_frame,
                                                                const int64_t* __restrict__ off, int64_t n_groups,
                                                                double* __restrict__ records, int debug_arg, MidSolveArgsT<T> sa) {
    // the timing switches (PDS_GMID_DEBUG) exist in development builds only: as a run-time value `debug & 8` put a scalar branch around
    // every matrix instruction but the first of a step and copies of the accumulators behind the unrolled half-tile loop
#ifdef PDS_DEV_SWITCHES
    const int debug = debug_arg;
#else
    constexpr int debug = 0;
    (void)debug_arg;
#endif
    constexpr int ES = (int)sizeof(T), EPL = 16 / ES;  // element bytes; elements per 16-byte lane piece
    static_assert(ES == 8 || PAIRED, "f32 frames: the paired form");
    static_assert(SPPC == 0 || NBLK == 2, "the in-wave solve serves two tile columns");
    static_assert(!PAIRED || SPPC > 0, "pairs exist for the in-kernel solve");
    static_assert(!PAIRED || MidPacked<SPPC ? SPPC : 1, YC>::COUNT * 8 <= (DIRECT ? kMidDirectSlotBytes : kMidSolveScratch), "the slot holds one group");
    static_assert(!YC || PAIRED, "the ones / target columns are the paired form's");
    static_assert(NQ == 0 || ((NQ == 1 || NQ == 2 || NQ == 3) && YC && NBLK == 2), "the quad form: ones and target inside the quads");
    static_assert(NQ != 3 || DIRECT, "the octet is the direct form's");
    static_assert(!DIRECT || (PAIRED && NBLK == 2 && NQ != 2 && (YC || (SPPC == 32 && ES == 8))) || (!PAIRED && NBLK == 4 && SPPC == 0 && !YC && NQ == 0 && ES == 8),
                  "the direct form: the paired kernels of 17 .. 32 features (the target beside the blocks: f64 frames), or the record stream of 33 .. 64");
    using MD = MidDims<NBLK, ES>;
    constexpr int IMG = DIRECT ? 0 : MD::LDS_BYTES;  // bytes of tile images in front of the pair's slot
    // (the direct form walks eight blocks of eight rows -- sixteen of f32 -- whatever the LDS form's half-tile of this NBLK is)
    constexpr int HR = DIRECT ? kMidDirectRows<ES> : MD::HR, GS = MD::GS, NPAIR = MD::NPAIR;
    extern __shared__ __attribute__((aligned(16))) char gmid_lds[];
    typedef __attribute__((address_space(3))) char* lds_c;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    typedef double mid_d2 __attribute__((ext_vector_type(2)));
    typedef volatile __attribute__((address_space(3))) unsigned* lds_flag;
#define PDS_GM_LDSD(addr) (*(__attribute__((address_space(3))) double*)(addr))
#define PDS_GM_LDST(addr) (*(__attribute__((address_space(3))) T*)(addr))
#ifdef PDS_PROFILE_MID
    unsigned long long mprof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto mprof_out = [&]() __attribute__((always_inline)) {
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 16; ++k)
                if (mprof[k]) atomicAdd(&g_mid_phase[k], mprof[k]);
    };
#endif
    PDS_MT(t_kernel);
    const int lane = threadIdx.x & 63;
    const int wv = PAIRED ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0, pairi = wv & 3;
    const bool consumer = PAIRED && wv >= 4;
    lds_c sm = (lds_c)gmid_lds + (PAIRED ? pairi * kMidPairLds<NBLK, DIRECT> : 0);
    constexpr int NSLOT = DIRECT ? (PAIRED ? kMidDirectSlots : 0) : 1, SLOTB = DIRECT ? kMidDirectSlotBytes : kMidSolveScratch;
    constexpr int DY_OFF = IMG + NSLOT * SLOTB;  // (DIRECT without YC: the target's two images behind the ring)
    const lds_flag FL = (lds_flag)(sm + DY_OFF + (DIRECT ? 2 * kMidDirectYBytes : 0));  // [0] trips published, [1] trips taken, [2] stream finished
    const int64_t wave = PAIRED ? (int64_t)blockIdx.x * 4 + pairi : (int64_t)blockIdx.x, nwaves = PAIRED ? (int64_t)gridDim.x * 4 : (int64_t)gridDim.x;
    const int64_t row_begin = off[0], row_end = off[n_groups];
    if (row_end <= row_begin) return;
    // half-tiles [H0, H1) cover the chunk's rows; a wave takes a contiguous range of them
    const int64_t H0 = row_begin / HR, H1 = (row_end + HR - 1) / HR;
    const int64_t h0 = H0 + (H1 - H0) * wave / nwaves, h1 = H0 + (H1 - H0) * (wave + 1) / nwaves;
    if (!consumer) {
        for (int i = lane * 16; i < IMG; i += 64 * 16) *(__attribute__((address_space(3))) mid_d2*)(sm + i) = mid_d2{0.0, 0.0};
        if constexpr (PAIRED)
            if (lane < 4) FL[lane] = 0u;
        if constexpr (YC && !DIRECT) {  // the ones column: the (otherwise unused) weight image of both half-tiles, written once
            PDS_WAVE_LDS_SYNC();
            for (int b = 0; b < MD::NBUF; ++b)
                for (int r = lane; r < HR; r += 64) PDS_GM_LDST(sm + b * MD::HALF_BYTES + MD::W_OFF + r * ES) = (T)1;
        }
        PDS_WAVE_LDS_SYNC();
    }
    if constexpr (PAIRED) __syncthreads();  // (the only workgroup barrier: the flag words are zero before a solver wave polls them)
    if (h0 >= h1) return;
    const int64_t W0 = h0 * HR > row_begin ? h0 * HR : row_begin, W1 = h1 * HR < row_end ? h1 * HR : row_end;  // the wave's rows
    const int q = p + 2;
    // SPPC: finished groups wait, up to four of them (one per 16-lane DPP row, two columns per lane: solve_row16_dev.hpp), and are
    // solved side by side.  A system next to the gate is marked; its record is rebuilt from the group's rows (a rare, slow path: the
    // accumulators it came from are gone by then).
    // pending systems: RAW moments (the centring, lambda and the diagonal wait for the solve, where they cost once per four groups)
    //   pa0 / pa1[i] = G[i][t] / G[i][16 + t], [SPPC] = X'y; psj0 / psj1 = column sums; pnn = rows; psy = sum y; pgid = group
    double pa0[SPPC + 1], pa1[SPPC + 1], psj0 = 0.0, psj1 = 0.0, pnn = 1.0, psy = 0.0;
    int64_t pgid = -1;
    int npend = 0;
    if constexpr (SPPC > 0) {
#pragma unroll
        for (int i = 0; i <= SPPC; ++i) pa0[i] = pa1[i] = 0.0;
    }
    auto record_from_rows = [&](int64_t gg, double* M) __attribute__((always_inline)) {
        const int64_t r0 = off[gg], r1 = off[gg + 1];
        for (int e = lane; e < q * q; e += 64) {
            const int i = e % q, j = e / q;
            if (i > j) continue;
            const gptr<T> ci = as_global(cols[i < p ? i : p]), cj = as_global(cols[j < p ? j : p]);  // (index p: the target column)
            double sacc = 0.0;
            for (int64_t r = r0; r < r1; ++r) {
                const double zi = i < p ? (double)ci[r] : (i == p ? 1.0 : (double)ci[r]);
                const double zj = j < p ? (double)cj[r] : (j == p ? 1.0 : (double)cj[r]);
                sacc = fma(zi, zj, sacc);
            }
            M[i + (int64_t)j * q] = sacc;
            M[j + (int64_t)i * q] = sacc;
        }
    };
    auto solve_pending = [&]() __attribute__((always_inline)) {
        if constexpr (SPPC > 0) {
            if (npend == 0) return;
            const int t = lane & 15, R = lane >> 4, pout = p + sa.sp.bias;
            const bool live = R < npend && pgid >= 0;  // (pgid < 0: padding of the paired form's last batch)
            // ---- raw moments -> the centred system with lambda on the diagonal (all pending rows at once)
            const bool c0v = t < p, c1v = 16 + t < p;
            pa0[SPPC] = c0v ? pa0[SPPC] : 0.0;
            pa1[SPPC] = c1v ? pa1[SPPC] : 0.0;
            double pdj0 = 1.0, pdj1 = 1.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {  // (selects, not a branch per i; a column beyond p gets a unit diagonal: its step changes nothing)
                const bool at = i == t;
                const double d0 = pa0[i] + sa.sp.lambda;
                pdj0 = (at && c0v) ? d0 : pdj0;
                pa0[i] = at ? (c0v ? d0 : 1.0) : pa0[i];
                if (16 + i < SPPC) {
                    const double d1 = pa1[16 + i] + sa.sp.lambda;
                    pdj1 = (at && c1v) ? d1 : pdj1;
                    pa1[16 + i] = at ? (c1v ? d1 : 1.0) : pa1[16 + i];
                }
            }
            if (!sa.sp.bias) psj0 = psj1 = 0.0;
            psj0 = c0v ? psj0 : 0.0;
            psj1 = c1v ? psj1 : 0.0;
            if (sa.sp.bias) {  // centre: G_ij - s_i s_j / n (the intercept never takes a lane)
                const double m0 = psj0 / pnn, m1 = psj1 / pnn;
                Row16Centre<SPPC, SPPC - 1>::run(pa0, pa1, psj0, psj1, m0, m1);
                pa0[SPPC] = fma(-psy, m0, pa0[SPPC]);
                pa1[SPPC] = fma(-psy, m1, pa1[SPPC]);
            }
            const bool pfew = pnn < (double)pout;  // "#Data < #features"
            double w0, w1;
            bool is_null, suspect;
            row16_ldl_solve<SPPC, (PAIRED && SPPC > 24)>(pa0, pa1, pdj0, pdj1, t, p, pfew, sa.sp, w0, w1, is_null, suspect);
            const double nanv = __builtin_nan("");
            const int64_t gq = live ? pgid : 0;
            T* co = sa.coeffs + gq * (int64_t)pout;
            if (live && t < p) co[t] = (T)(is_null ? nanv : w0);
            if (live && 16 + t < p) co[16 + t] = (T)(is_null ? nanv : w1);
            if (sa.sp.bias) {
                const double sb = Grp<16>::sum((t < p ? psj0 * w0 : 0.0) + (16 + t < p ? psj1 * w1 : 0.0));
                if (live && t == 0) co[p] = (T)(is_null ? nanv : (psy - sb) / pnn);
            }
            if (live && t == 0) sa.flags[gq] = is_null ? 1 : 0;
            // marked systems: one DPP row after the other (wave-uniform control flow)
            const unsigned long long sus = __builtin_amdgcn_ballot_w64(live && suspect);
            for (int r = 0; r < 4; ++r) {
                if (!((sus >> (16 * r)) & 1ull)) continue;
                const int glo = __builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)pgid, 16 * r);
                const int ghi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)pgid >> 32), 16 * r);
                const int64_t gg = (int64_t)(((uint64_t)(uint32_t)ghi << 32) | (uint64_t)(uint32_t)glo);
                unsigned slot = 0;
                if (lane == 0) slot = atomicAdd(sa.mark_count, 1u);
                slot = (unsigned)__builtin_amdgcn_readfirstlane((int)slot);
                if (slot < sa.mark_cap) {
                    if (lane == 0) sa.mark_list[slot] = (int32_t)gg;
                    record_from_rows(gg, sa.mark_rec + (int64_t)slot * q * q);
                }
            }
            npend = 0;
        }
    };
    // PAIRED, solving wave: a published slot -> DPP row `npend` of the pending registers (row-masked moves, as route_pending), four
    // pending -> solve.  The slot holds the UPPER TRIANGLE of G row by row (MidPacked: one trip at any width; the full columns of
    // route_pending's layout took two from 25 features, and the streaming wave stood still between them while a solve ran): lane t
    // reads row i of its column at U(i, t) for i <= t and at U(t, i) below the diagonal -- an immediate offset on one of two lane
    // addresses, chosen per element only inside the two diagonal blocks.
    auto solver_wave = [&]() __attribute__((always_inline)) {
        if constexpr (PAIRED) {
            typedef __attribute__((address_space(3))) double* lds_dp;
            using PK = MidPacked<SPPC, YC>;
            const int t = lane & 15;
            const bool c1 = YC ? 16 + t < p : 16 + t < SPPC;  // (YC: columns p, p + 1 of the slot are the ones and the target, not the system's)
            const int u = c1 ? 16 + t : 16;  // (lanes without a second column read a valid address, their values are zeroed)
            const int rowt = PK::tri_rt(t) - t, rowu = PK::tri_rt(u) - u;  // U(t, i) = S[rowt + i], U(u, i) = S[rowu + i]
            unsigned cseq = 0;
            // A batch is straight-line code: group 0 goes to EVERY row (plain reads: the registers are defined afresh, nothing of the last
            // batch stays live), groups 1 .. 3 into their rows by row-masked DPP moves, then the solve.  With the row as a run-time
            // value (a switch over four variants, or a branch on the lane's row) every pending register existed twice around the merge
            // points -- copies, and a register file that spilled into the stream's memory queue: 15 000 clk per group.
            auto wait_group = [&]() __attribute__((always_inline)) {  // false: the stream has finished and nothing is published
                for (;;) {
                    const unsigned d = FL[2], f = FL[0];  // (in this order: after `done` nothing is published)
                    if (f != cseq) return true;
                    if (d) return false;
                    __builtin_amdgcn_s_sleep(1);
                }
            };
            auto took = [&]() __attribute__((always_inline)) {
                PDS_WAVE_LDS_SYNC();  // (the values are in registers)
                cseq += DIRECT ? 4 : 1;
                FL[1] = cseq;
            };
            auto take = [&](auto rm) __attribute__((always_inline)) {
                constexpr int ROW = decltype(rm)::value;
                // DIRECT: FOUR groups at once -- row R of the wave reads slot (cseq + R) mod NSLOT (cseq is a multiple of four there), every
                // lane's read lands in its own row: no row-masked moves, a quarter of the reads per group
                const lds_dp S = (lds_dp)(sm + IMG + (DIRECT ? ((int)(cseq & (NSLOT - 1)) + (lane >> 4)) * SLOTB : 0));
                auto put = [&](double& dst, double v) __attribute__((always_inline)) {
                    if constexpr (ROW == 0) dst = v;
                    else dst = __builtin_amdgcn_update_dpp(dst, v, 0xE4 /*quad_perm:[0,1,2,3]*/, 1 << ROW, 0xf, false);
                };
                // (the lane terms pass through an empty asm: per-lane addresses inside the diagonal blocks are loop invariants otherwise,
                // one register each)
                int tl = t, ul = u, rt = rowt, ru = rowu;
                asm volatile("" : "+v"(tl), "+v"(ul), "+v"(rt), "+v"(ru));
                // eight rows (sixteen reads) at a time, the next eight in flight while these are moved into the pending registers
                constexpr int NB = SPPC / 8;
                double va[2][8], vb[2][8];
                auto fetch8 = [&](int b, double (&x0)[8], double (&x1)[8]) __attribute__((always_inline)) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int i = 8 * b + k;
                        // column t: rows beyond 15 are always below the diagonal; column 16 + t: rows up to 15 are always above it
                        const int e0 = (i < 16 && i <= t) ? PK::tri(i) - i + tl : rt + i;
                        const int e1 = (i < 16 || i <= u) ? PK::tri(i) - i + ul : ru + i;
                        x0[k] = S[e0];
                        x1[k] = S[e1];
                    }
                };
                auto put8 = [&](int b, const double (&x0)[8], const double (&x1)[8]) __attribute__((always_inline)) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int i = 8 * b + k;
                        const bool rowv = !YC || i < 16 || i < p;  // (YC: rows p, p + 1 of the slot are not the system's either)
                        put(pa0[i], rowv ? x0[k] : 0.0);
                        put(pa1[i], (c1 && rowv) ? x1[k] : 0.0);
                    }
                };
                fetch8(0, va[0], vb[0]);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (b + 1 < NB) fetch8(b + 1, va[(b + 1) & 1], vb[(b + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    put8(b, va[b & 1], vb[b & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // X'y, column sums, rows, sum y -- YC: entries (., p + 1), (., p), (p, p), (p, p + 1) of the triangle
                const double x0 = YC ? S[rt + p + 1] : S[PK::XY + tl], x1 = YC ? S[ru + p + 1] : S[PK::XY + ul];
                const double s0 = YC ? S[rt + p] : S[PK::CS + tl], s1 = YC ? S[ru + p] : S[PK::CS + ul];
                put(pa0[SPPC], x0);
                put(pa1[SPPC], c1 ? x1 : 0.0);
                put(psj0, s0);
                put(psj1, c1 ? s1 : 0.0);
                put(pnn, YC ? S[PK::tri_rt(p)] : S[PK::TAIL]);
                put(psy, YC ? S[PK::tri_rt(p) + 1] : S[PK::TAIL + 1]);
                {
                    const long long gv = __double_as_longlong(S[PK::GID]);
                    if constexpr (ROW == 0) {
                        pgid = (int64_t)gv;
                    } else {
                        int lo = (int)(uint32_t)(uint64_t)pgid, hi = (int)(uint32_t)((uint64_t)pgid >> 32);
                        lo = __builtin_amdgcn_update_dpp(lo, (int)(uint32_t)(uint64_t)gv, 0xE4, 1 << ROW, 0xf, false);
                        hi = __builtin_amdgcn_update_dpp(hi, (int)(uint32_t)((uint64_t)gv >> 32), 0xE4, 1 << ROW, 0xf, false);
                        pgid = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
                    }
                }
                took();
            };
            // (the streaming wave pads its last batch with discarded groups -- gid < 0 -- so a batch is always four: the only way out of the
            // loop is in front of a batch, with no pending register live)
            auto next_group = [&]() __attribute__((always_inline)) {
                while (FL[0] == cseq) __builtin_amdgcn_s_sleep(1);
            };
            if constexpr (DIRECT) {
                for (;;) {
                    PDS_MT(c0);
                    bool more = true;
                    for (;;) {  // four groups published (the streaming wave pads its last batch), or the stream finished with nothing left
                        const unsigned d = FL[2], f = FL[0];  // (in this order: after `done` nothing is published)
                        if ((unsigned)(f - cseq) >= 4u) break;
                        if (d) {
                            more = false;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    PDS_MADD(8, c0);
                    if (!more) break;
                    PDS_MT(c1);
                    take(std::integral_constant<int, 0>{});
                    PDS_MADD(9, c1);
                    npend = 4;
                    PDS_MT(c8);
                    if (!(debug & 4)) solve_pending();
                    PDS_MADD(10, c8);
                }
                return;
            }
            for (;;) {
                PDS_MT(c0);
                if (!wait_group()) break;
                PDS_MADD(8, c0);
                PDS_MT(c1);
                take(std::integral_constant<int, 0>{});
                PDS_MADD(9, c1);
                PDS_MT(c2);
                next_group();
                PDS_MADD(8, c2);
                PDS_MT(c3);
                take(std::integral_constant<int, 1>{});
                PDS_MADD(9, c3);
                PDS_MT(c4);
                next_group();
                PDS_MADD(8, c4);
                PDS_MT(c5);
                take(std::integral_constant<int, 2>{});
                PDS_MADD(9, c5);
                PDS_MT(c6);
                next_group();
                PDS_MADD(8, c6);
                PDS_MT(c7);
                take(std::integral_constant<int, 3>{});
                PDS_MADD(9, c7);
                npend = 4;
                PDS_MT(c8);
                if (!(debug & 4)) solve_pending();  // (bit 4, a timing experiment: routed, never solved)
                PDS_MADD(10, c8);
            }
        }
    };
    if constexpr (PAIRED)
        if (consumer) {  // (before the streaming wave's state exists: none of it is live in the solver's registers)
            if (debug & 16) return;  // (timing experiment: no solving wave at all, nothing published)
            solver_wave();
#ifdef PDS_PROFILE_MID
            PDS_MADD(15, t_kernel);
            mprof_out();
#endif
            return;
        }
    const int g_ = lane / MD::GL, piece = lane % MD::GL;
    const T* cbase[16];
    unsigned valid = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = 16 * g_ + i;
        cbase[i] = cols[c < p ? c : p] + EPL * piece;
        if (c < p) valid |= 1u << i;
    }
    const T* ybase = cols[p] + EPL * lane;
    auto issue = [&](int buf, int64_t row0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if ((valid >> i) & 1u)
                __builtin_amdgcn_global_load_lds((glb_ptr)(as_global(cbase[i]) + row0), (lds_ptr)(sm + buf * MD::HALF_BYTES + i * GS), 16, 0, 0);
        if (lane < HR / EPL)
            __builtin_amdgcn_global_load_lds((glb_ptr)(as_global(ybase) + row0), (lds_ptr)(sm + buf * MD::HALF_BYTES + MD::Y_OFF), 16, 0, 0);
    };
    auto load_guarded = [&](int buf, int64_t row0) __attribute__((always_inline)) {  // the frame's last, partial half-tile
        for (int c = 0; c <= p; ++c) {
            const int offb = c < p ? (c % 16) * GS + (c / 16) * HR * ES : MD::Y_OFF;
            const gptr<T> col = as_global(cols[c]);
            for (int r = lane; r < HR; r += 64) PDS_GM_LDST(sm + buf * MD::HALF_BYTES + offb + r * ES) = row0 + r < n_frame ? col[row0 + r] : (T)0;
        }
    };
    auto fetch_tile = [&](int buf, int64_t h) __attribute__((always_inline)) {
        if ((h + 1) * HR <= n_frame) issue(buf, h * HR);
        else load_guarded(buf, h * HR);
    };
    d4 acc[NPAIR];
    double xy[NBLK], cs[NBLK], yy = 0.0, ys = 0.0;
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NPAIR; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int b = 0; b < NBLK; ++b) xy[b] = cs[b] = 0.0;
        yy = ys = 0.0;
    };
    zero_acc();
    const int fi = lane & 15, fk = lane >> 4;
    // NQ = 3 (direct form, 19 .. 22 features): the second piece is an OCTET -- columns 16 .. 23 (x16 .., the ones, the target), lane fi holds
    // column 16 + fi % 8, so the four lanes of block b hold quad b % 2 of it; rotated by four lanes inside the row (one DPP move per half)
    // they hold the other quad.  Four v_mfma_f64_4x4x4_4b per step -- first block x octet as it is and rotated, octet x octet likewise -- give
    // the first block's products with all eight columns and the 8 x 8 corner: 64 ticks of the matrix pipe where two 16 x 16 x 4 took 128.
    constexpr int NOP = NQ == 3 ? 2 : (NQ ? 1 + NQ : NBLK);  // operand registers per step: the first block + NQ quads (or the octet), or the NBLK blocks
    int opo[NOP];  // the lane's operand column inside an image
#pragma unroll
    for (int b = 0; b < NOP; ++b) {
        const int c = NQ == 3 ? (b == 0 ? fi : 16 + (fi & 7)) : NQ ? (b == 0 ? fi : 12 + 4 * b + (fi & 3)) : 16 * b + fi;  // (quad b - 1: column 16 + 4 (b - 1) + lane % 4)
        opo[b] = (c & 15) * GS + (c >> 4) * HR * ES;
        if constexpr (YC) {  // columns p and p + 1: the ones image and the target's image
            if (b > 0 && c == p) opo[b] = MD::W_OFF;
            if (b > 0 && c == p + 1) opo[b] = MD::Y_OFF;
        }
    }
    const int qb = (lane >> 2) & 3, qj = lane & 3;  // quad lanes: D[i = fk][j = qj] of block qb
    // ---- DIRECT: two register sets of NB8 blocks x NOP pieces x 16 bytes; the lane's address per piece moves on by one half-tile per issue
    // (f32 frames: the same 16 bytes per lane are FOUR rows -- blocks of 16 rows, four steps each, the operands widened as they are multiplied)
    constexpr int RPL = EPL, BR = 4 * RPL, BSH = ES == 8 ? 3 : 4, NB8 = HR / BR;
    static_assert(!DIRECT || NB8 == 8, "eight blocks per half-tile");
    using VT = typename Tile<T>::vec;
    VT CUR[DIRECT ? NB8 : 1][NOP], NXT[DIRECT ? NB8 : 1][NOP];
    gptr<char> dcp[NOP];
    bool dreal[NOP];
    if constexpr (DIRECT) {
#pragma unroll
        for (int b = 0; b < NOP; ++b) {
            const int c = NQ == 3 ? (b == 0 ? fi : 16 + (fi & 7)) : NQ ? (b == 0 ? fi : 12 + 4 * b + (fi & 3)) : 16 * b + fi;  // (as opo: the piece's column in this lane)
            const bool real = c < p || (YC && c == p + 1);                           // a frame column (YC: p + 1 = the target, p = ones); beyond: zeros
            const char* cst = ES == 8 ? reinterpret_cast<const char*>(g_mid_direct_const) : reinterpret_cast<const char*>(g_mid_direct_const_f32);
            const char* base = real ? reinterpret_cast<const char*>(cols[c < p ? c : p]) + h0 * (int64_t)(HR * ES) : cst + ((YC && c == p) ? 0 : HR * ES);
            dcp[b] = (gptr<char>)(base + 16 * fk);
            dreal[b] = real;
        }
    }
    // (Measured and not kept, profiles/r06_grouped_mid_direct.txt: the next half-tile's loads dealt out between the blocks of this one --
    // inline asm loads, the wait at the top of the next half-tile: the last blocks' loads have no time to land, 17 features 3.42 -> 4.25 ms;
    // ONE register set refilled in place with a wait in front of every block: the compiler keeps such a set in two places and moves it
    // between them wherever it likes -- a move of a register whose load is in flight reads what was there before.)
    auto advance_direct = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < NOP; ++b)
            if (b == 0 || dreal[b]) dcp[b] += HR * ES;  // (the first block's columns are all frame columns)
    };
    auto issue_direct = [&](int64_t h) __attribute__((always_inline)) {
        if constexpr (DIRECT) {
            if constexpr (!YC) {  // 31 / 32 features: the target rides beside the blocks -- its image of this half-tile in LDS, read per step
                                  // (requested FIRST: behind the sixteen register loads the one asynchronous piece cost the wave 600 clk)
                const lds_c yimg = sm + DY_OFF + (int)((h - h0) & 1) * kMidDirectYBytes;
                if ((h + 1) * HR <= n_frame) {
                    if (lane < HR / EPL) __builtin_amdgcn_global_load_lds((glb_ptr)(as_global(ybase) + h * HR), (lds_ptr)yimg, 16, 0, 0);
                } else {
                    const gptr<T> col = as_global(cols[p]);
                    for (int r = lane; r < HR; r += 64) PDS_GM_LDST(yimg + r * ES) = h * HR + r < n_frame ? col[h * HR + r] : (T)0;
                }
            }
            if ((h + 1) * HR <= n_frame) {
#pragma unroll
                for (int k = 0; k < NB8; ++k)
#pragma unroll
                    for (int b = 0; b < NOP; ++b) NXT[k][b] = PDS_MID_DIRECT_LOAD(reinterpret_cast<gptr<VT>>(dcp[b] + 64 * k));
            } else {  // the frame's last, partial half-tile: row by row, rows beyond the frame are zeros (nobody multiplies them)
#pragma unroll
                for (int k = 0; k < NB8; ++k)
#pragma unroll
                    for (int b = 0; b < NOP; ++b)
#pragma unroll
                        for (int j = 0; j < RPL; ++j) {
                            const int64_t row = h * HR + BR * k + RPL * fk + j;
                            T v = (T)0;
                            if (!dreal[b] || row < n_frame) v = *reinterpret_cast<gptr<T>>(dcp[b] + 64 * k + ES * j);
                            NXT[k][b][j] = v;
                        }
            }
            advance_direct();
        }
    };
    // ---- the group that holds the wave's first row
    int64_t g = 0;
    {
        int64_t lo = 0, hi = n_groups;  // last g with off[g] <= W0
        while (hi - lo > 1) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (off[mid] <= W0) lo = mid;
            else hi = mid;
        }
        g = lo;
    }
    // the group walk is wave-uniform: with g known to be uniform the offsets come through the SCALAR cache (their own counter) --
    // as vector loads every group end waited, through the in-order vmcnt, for the next half-tile's 17 loads as well (12 us per
    // half-tile instead of 2); the end of the NEXT group is fetched one group ahead
    auto uni64 = [](int64_t v) __attribute__((always_inline)) {
        const int lo32 = __builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)v), hi32 = __builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
        return (int64_t)(((uint64_t)(uint32_t)hi32 << 32) | (uint64_t)(uint32_t)lo32);
    };
    g = uni64(g);
    int64_t gs = off[g], ge = off[g + 1];
    int64_t ge_next = off[g + 2 <= n_groups ? g + 2 : n_groups];
    int rows_in_acc = 0;  // (rows of the open group in THIS wave's accumulators: a wave's range is far below 2^31 rows)
    // accumulate rows [lo, hi) (relative to the half-tile in `buf`) into acc: a masked step at either end where the segment does not
    // start / end on a 4-row step, the steps in between unmasked with the next step's operands fetched from LDS before this step
    // multiplies (fetch -> wait -> multiply per step ran at half the stream rate: one wave per SIMD, nobody else hides the round trip)
    auto consume = [&](int buf, int lo, int hi) __attribute__((always_inline)) {
        const lds_c base = sm + buf * MD::HALF_BYTES;
        auto fetch = [&](int s, double (&a)[NOP], double& yk) __attribute__((always_inline)) {
            const int roff = (4 * s + fk) * ES;
#pragma unroll
            for (int b = 0; b < NOP; ++b) a[b] = (double)PDS_GM_LDST(base + opo[b] + roff);
            if constexpr (!YC) yk = (double)PDS_GM_LDST(base + MD::Y_OFF + roff);
            else yk = 0.0;
        };
        auto mult = [&](const double (&a)[NOP], double yk) __attribute__((always_inline)) {
            if constexpr (NQ == 1) {
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], a[0], acc[0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], a[1], acc[1][0], 0, 0, 0);
                acc[2][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], a[1], acc[2][0], 0, 0, 0);
            } else if constexpr (NQ == 3) {
                const int lo32 = __double2loint(a[1]), hi32 = __double2hiint(a[1]);  // the octet, rotated by four lanes inside its row
                const double ar = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi32, 0x124 /*row_ror:4*/, 0xf, 0xf, true),
                                                   __builtin_amdgcn_update_dpp(0, lo32, 0x124, 0xf, 0xf, true));
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], a[0], acc[0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], a[1], acc[1][0], 0, 0, 0);  // G[4 qb + fk][16 + 4 (qb % 2) + qj]
                acc[1][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], ar, acc[1][1], 0, 0, 0);    // G[4 qb + fk][16 + 4 (1 - qb % 2) + qj]
                acc[1][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], a[1], acc[1][2], 0, 0, 0);  // G[16 + 4 (qb % 2) + fk][16 + 4 (qb % 2) + qj]
                acc[1][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], ar, acc[1][3], 0, 0, 0);    // G[16 + 4 (qb % 2) + fk][16 + 4 (1 - qb % 2) + qj]
            } else if constexpr (NQ == 2) {
                // the 8 x 8 corner in ONE instruction: block 0 = quad 0 with itself, 1 = quad 0 with quad 1, 2 (and 3) = quad 1 with itself
                const double ca = qb < 2 ? a[1] : a[2], cb = qb == 0 ? a[1] : a[2];
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], a[0], acc[0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], a[1], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], a[2], acc[1][1], 0, 0, 0);
                acc[2][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(ca, cb, acc[2][0], 0, 0, 0);
            } else {
                int t = 0;
#pragma unroll
                for (int I = 0; I < NBLK; ++I)
#pragma unroll
                    for (int J = I; J < NBLK; ++J) {
                        if (t == 0 || !(debug & 8))  // (timing experiment: the first block only)
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], a[J], acc[t], 0, 0, 0);
                        ++t;
                    }
            }
            if constexpr (!YC) {
#pragma unroll
                for (int b = 0; b < NBLK; ++b) {
                    xy[b] = fma(a[b], yk, xy[b]);
                    cs[b] += a[b];
                }
                yy = fma(yk, yk, yy);
                ys += yk;
            }
        };
        if constexpr (DIRECT) {
            // Blocks of eight rows; a lane's rows in block kb are 8 kb + 2 fk + j, step j of the block multiplies row j of every lane.
            const lds_c yrow = sm + DY_OFF + buf * kMidDirectYBytes + 16 * fk;  // (without YC: the lane's two target values of block kb at + 64 kb)
            auto two_steps = [&](const VT (&c)[NOP], mid_d2 yv) __attribute__((always_inline)) {  // (f32 frames: four)
#pragma unroll
                for (int j = 0; j < RPL; ++j) {
                    double a[NOP];
#pragma unroll
                    for (int b = 0; b < NOP; ++b) a[b] = (double)c[b][j];
                    mult(a, YC ? 0.0 : yv[j & 1]);
                }
            };
            auto y_of = [&](int kb) __attribute__((always_inline)) {
                if constexpr (YC) return mid_d2{0.0, 0.0};
                else return *(__attribute__((address_space(3))) mid_d2*)(yrow + 64 * kb);
            };
            // A block the segment covers in part (at most one at either end) is copied out -- a switch over the block index: moves of
            // operand registers only --, the rows outside the segment zeroed, and multiplied: ONE copy of that code in a loop of two.  The
            // blocks it covers as a whole, [f0, f1), are multiplied straight from their registers by NB8 predicated copies of the two steps
            // (if-then without else: the accumulators stay where they are on either path.  A run of the same copies entered by a switch
            // and left by `break`s came back from the compiler with every accumulator moved between two register sets at each block
            // boundary, behind an s_nop that drains the matrix pipe: 170 instead of 100 clk per step.)
            auto partial = [&](int kb) __attribute__((always_inline)) {
                VT c8[NOP];
#define PDS_GM_CASE(K)                                                                       \
    case K:                                                                                  \
        _Pragma("unroll") for (int b = 0; b < NOP; ++b) c8[b] = CUR[K < NB8 ? K : 0][b];    \
        break;
                switch (kb) {
                    PDS_GM_CASE(0) PDS_GM_CASE(1) PDS_GM_CASE(2) PDS_GM_CASE(3) PDS_GM_CASE(4) PDS_GM_CASE(5) PDS_GM_CASE(6)
                    default:
#pragma unroll
                        for (int b = 0; b < NOP; ++b) c8[b] = CUR[NB8 - 1][b];
                        break;
                }
#undef PDS_GM_CASE
                const int rr = BR * kb + RPL * fk;
                const bool in0 = rr >= lo && rr < hi, in1 = rr + 1 >= lo && rr + 1 < hi;
#pragma unroll
                for (int j = 0; j < RPL; ++j) {
                    const bool in = rr + j >= lo && rr + j < hi;
#pragma unroll
                    for (int b = 0; b < NOP; ++b) c8[b][j] = in ? c8[b][j] : (T)0;
                }
                mid_d2 yv = y_of(kb);
                yv[0] = in0 ? yv[0] : 0.0;
                yv[1] = in1 ? yv[1] : 0.0;
                two_steps(c8, yv);
            };
            const int f0 = (lo + BR - 1) >> BSH, f1 = hi >> BSH;
            const int pk0 = (f0 > f1 || (lo & (BR - 1))) ? lo >> BSH : ((hi & (BR - 1)) ? hi >> BSH : -1);
            const int pk1 = (f0 <= f1 && (lo & (BR - 1)) && (hi & (BR - 1))) ? hi >> BSH : -1;
            PDS_MT(tp13);
            for (int t = 0; t < 2; ++t) {
                const int kb = t == 0 ? pk0 : pk1;
                if (kb < 0) break;
                partial(kb);
            }
            PDS_MADD(13, tp13);
            PDS_MT(tp12);
#pragma unroll
            for (int K = 0; K < NB8; ++K)
                if (K >= f0 && K < f1) two_steps(CUR[K], y_of(K));
            PDS_MADD(12, tp12);
            rows_in_acc += hi - lo;
            return;
        }
        auto masked = [&](int s) __attribute__((always_inline)) {
            const int rr = 4 * s + fk;
            const bool in = rr >= lo && rr < hi;
            double a[NOP], yk;
            fetch(s, a, yk);
#pragma unroll
            for (int b = 0; b < NOP; ++b) a[b] = in ? a[b] : 0.0;
            yk = in ? yk : 0.0;
            mult(a, yk);
        };
        // a whole half-tile of one group: the unrolled form of the single-regression kernel (not for f32 frames with the side sums: its 32
        // steps cost that variant 65 spilled registers -- the column pointers, reloaded behind the loads in flight)
        if ((ES == 8 || YC) && lo == 0 && hi == HR) {
            double a[NOP], yk;
            fetch(0, a, yk);
#pragma unroll
            for (int s = 0; s < MD::NS; ++s) {
                double an[NOP], ykn = 0.0;
#pragma unroll
                for (int b = 0; b < NOP; ++b) an[b] = 0.0;
                if (s + 1 < MD::NS) fetch(s + 1, an, ykn);
                mult(a, yk);
#pragma unroll
                for (int b = 0; b < NOP; ++b) a[b] = an[b];
                yk = ykn;
            }
            rows_in_acc += HR;
            return;
        }
        int s0 = lo >> 2, s1 = (hi + 3) >> 2;  // steps [s0, s1)
        if ((lo & 3) != 0 || s1 - s0 == 1) {    // (a segment inside one step: that step masked on both sides)
            masked(s0);
            ++s0;
        }
        if (s0 < s1 && (hi & 3) != 0) {
            --s1;
            masked(s1);
        }
        if (s0 < s1) {
            double a[NOP], yk;
            fetch(s0, a, yk);
            for (int s = s0; s < s1; ++s) {
                double an[NOP], ykn;
                fetch(s + 1 < s1 ? s + 1 : s, an, ykn);
                mult(a, yk);
#pragma unroll
                for (int b = 0; b < NOP; ++b) a[b] = an[b];
                yk = ykn;
            }
        }
        rows_in_acc += hi - lo;
    };
    // the accumulated rows of group g -> a record at M (plain stores, or atomics into a zeroed record that other waves add to)
    // (DIRECT: the lane terms of the record / slot addresses pass through an empty asm -- as loop invariants every one of them held a
    // register for the whole stream, and the streaming wave's two operand sets leave none: spilled, they were reloaded in front of
    // `s_waitcnt vmcnt(0)` behind the next half-tile's loads.  The 17 / 18-feature f64 kernel has the registers: hoisted there, 3.40 -> 3.32 ms;
    // the octet kernel has them too and is 3 % SLOWER with the addresses hoisted)
    auto put_record = [&](double* M, bool plain) __attribute__((always_inline)) {
        int l0 = fi, l1 = fk, l2 = qb, l3 = qj;
        if constexpr (DIRECT && (SPPC == 32 || ES == 4 || NQ != 1 || PDS_MID_LAUNDER_ALL)) asm volatile("" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));
        const int fi = l0, fk = l1, qb = l2, qj = l3;
        auto put = [&](int64_t idx, double v) __attribute__((always_inline)) {
            if (plain) M[idx] = v;
            else if (v != 0.0) unsafeAtomicAdd(M + idx, v);
        };
        int t = 0;
        const int pe = YC ? q : p;  // (YC: the blocks hold [X 1 y]' [X 1 y], which is the record)
        if constexpr (NQ == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = fk + 4 * r, j = fi;
                if (i < pe && j < pe) put(i + (int64_t)j * q, acc[0][r]);
            }
            const int par = qb & 1, ri = 4 * qb + fk, c0 = 16 + 4 * par + qj, c1 = 16 + 4 * (par ^ 1) + qj;
            if (c0 < pe) {
                put(ri + (int64_t)c0 * q, acc[1][0]);
                put(c0 + (int64_t)ri * q, acc[1][0]);
            }
            if (c1 < pe) {
                put(ri + (int64_t)c1 * q, acc[1][1]);
                put(c1 + (int64_t)ri * q, acc[1][1]);
            }
            const int rr = 16 + 4 * par + fk;
            if (qb < 2 && rr < pe) {  // (blocks 2, 3 repeat blocks 0, 1)
                if (c0 < pe) put(rr + (int64_t)c0 * q, acc[1][2]);  // the diagonal 4 x 4 blocks: both triangles in these lanes
                if (c1 < pe) put(rr + (int64_t)c1 * q, acc[1][3]);  // block 0: rows 16 .., columns 20 ..; block 1: its transpose
            }
            return;
        }
        if constexpr (NQ != 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = fk + 4 * r, j = fi;
                if (i < pe && j < pe) put(i + (int64_t)j * q, acc[0][r]);
            }
#pragma unroll
            for (int u = 0; u < NQ; ++u) {  // first block x quad u: G[4 qb + fk][16 + 4 u + qj]
                const int c = 16 + 4 * u + qj;
                if (c < pe) {
                    put((4 * qb + fk) + (int64_t)c * q, acc[1][u]);
                    put(c + (int64_t)(4 * qb + fk) * q, acc[1][u]);
                }
            }
            {  // the corner: NQ = 1: block 0 = (quad 0, quad 0), both triangles in its lanes; NQ = 2: blocks (0, 0), (0, 1), (1, 1)
                const int ri = 16 + (NQ == 2 && qb >= 2 ? 4 : 0) + fk, cj = 16 + (NQ == 2 && qb >= 1 ? 4 : 0) + qj;
                if (qb < (NQ == 2 ? 3 : 1) && ri < pe && cj < pe) {
                    put(ri + (int64_t)cj * q, acc[2][0]);
                    if (NQ == 2 && qb == 1) put(cj + (int64_t)ri * q, acc[2][0]);
                }
            }
            return;
        }
#pragma unroll
        for (int I = 0; I < NBLK; ++I)
#pragma unroll
            for (int J = I; J < NBLK; ++J) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * I + fk + 4 * r, j = 16 * J + fi;
                    if (i < pe && j < pe) {  // (j runs with the lane: G[j][i] first -- sixteen lanes, one line; the tile's own position is the strided one)
                        put(j + (int64_t)i * q, acc[t][r]);
                        if (I != J) put(i + (int64_t)j * q, acc[t][r]);
                    }
                }
                ++t;
            }
        if constexpr (YC) return;
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            double vx = xy[b], vc = cs[b];
            vx += __shfl_xor(vx, 16);
            vx += __shfl_xor(vx, 32);
            vc += __shfl_xor(vc, 16);
            vc += __shfl_xor(vc, 32);
            const int f = 16 * b + fi;
            if (fk == 0 && f < p) {
                put(f + (int64_t)(p + 1) * q, vx);
                put((p + 1) + (int64_t)f * q, vx);
                put(f + (int64_t)p * q, vc);
                put(p + (int64_t)f * q, vc);
            }
        }
        double vyy = yy, vys = ys;
        vyy += __shfl_xor(vyy, 16);
        vyy += __shfl_xor(vyy, 32);
        vys += __shfl_xor(vys, 16);
        vys += __shfl_xor(vys, 32);
        if (lane == 0) {
            put(p + (int64_t)p * q, (double)rows_in_acc);
            put(p + (int64_t)(p + 1) * q, vys);
            put((p + 1) + (int64_t)p * q, vys);
            put((p + 1) + (int64_t)(p + 1) * q, vyy);
        }
    };
    // the finished group's accumulators -> DPP row `npend` of the pending registers, through the LDS scratch behind the tile images
    // (every lane of every row reads column t / 16 + t; a ROW-MASKED DPP move -- identity permutation, row_mask = the pending row --
    // drops the values into that row's lanes only: two instructions per value, no select against the old contents, no temporaries)
    auto route_pending = [&]() __attribute__((always_inline)) {
        if constexpr (SPPC > 0) {
            typedef __attribute__((address_space(3))) double* lds_dp;
            constexpr int SS = SPPC + 2;                                   // doubles per column of the scratch
            constexpr bool ONE_TRIP = SPPC * SS * 8 <= kMidSolveScratch;   // all SPPC columns at once (up to 24 features)
            lds_dp S = (lds_dp)(sm + IMG);
            const int t = lane & 15;
            double vx0 = xy[0], vx1 = xy[1], vc0 = cs[0], vc1 = cs[1], vys = ys;
            vx0 += __shfl_xor(vx0, 16); vx0 += __shfl_xor(vx0, 32);
            vx1 += __shfl_xor(vx1, 16); vx1 += __shfl_xor(vx1, 32);
            vc0 += __shfl_xor(vc0, 16); vc0 += __shfl_xor(vc0, 32);
            vc1 += __shfl_xor(vc1, 16); vc1 += __shfl_xor(vc1, 32);
            vys += __shfl_xor(vys, 16); vys += __shfl_xor(vys, 32);
            const double nn = (double)rows_in_acc;
            auto into_row = [&](auto rm) __attribute__((always_inline)) {
                constexpr int RM = 1 << decltype(rm)::value;
                auto put = [&](double& dst, double v) __attribute__((always_inline)) {
                    dst = __builtin_amdgcn_update_dpp(dst, v, 0xE4 /*quad_perm:[0,1,2,3]*/, RM, 0xf, false);
                };
                auto put64 = [&](int64_t& dst, int64_t v) __attribute__((always_inline)) {
                    int lo = (int)(uint32_t)(uint64_t)dst, hi = (int)(uint32_t)((uint64_t)dst >> 32);
                    lo = __builtin_amdgcn_update_dpp(lo, (int)(uint32_t)(uint64_t)v, 0xE4, RM, 0xf, false);
                    hi = __builtin_amdgcn_update_dpp(hi, (int)(uint32_t)((uint64_t)v >> 32), 0xE4, RM, 0xf, false);
                    dst = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
                };
                PDS_WAVE_LDS_SYNC();
                // columns 0 .. 15: rows 0 .. 15 from block (0, 0), rows 16 .. from block (0, 1) transposed (G is symmetric)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    S[fi * SS + fk + 4 * r] = acc[0][r];
                    if (16 + fi < SPPC) S[(fk + 4 * r) * SS + 16 + fi] = acc[1][r];
                }
                if constexpr (ONE_TRIP) {
                    // columns 16 .. SPPC - 1 behind them: rows 0 .. 15 from block (0, 1), rows 16 .. from block (1, 1)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (16 + fi < SPPC) {
                            S[(16 + fi) * SS + fk + 4 * r] = acc[1][r];
                            if (16 + fk + 4 * r < SPPC) S[(16 + fi) * SS + 16 + fk + 4 * r] = acc[2][r];
                        }
                    }
                }
                PDS_WAVE_LDS_SYNC();
#pragma unroll
                for (int i = 0; i < SPPC; ++i) put(pa0[i], S[t * SS + i]);
                if constexpr (ONE_TRIP) {
                    const int t1 = (16 + t < SPPC) ? 16 + t : 0;  // (lanes without a second column read a valid address, their values are zeroed)
#pragma unroll
                    for (int i = 0; i < SPPC; ++i) {
                        const double v = S[t1 * SS + i];
                        put(pa1[i], (16 + t < SPPC) ? v : 0.0);
                    }
                } else {
                    PDS_WAVE_LDS_SYNC();
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        S[fi * SS + fk + 4 * r] = acc[1][r];
                        S[fi * SS + 16 + fk + 4 * r] = acc[2][r];
                    }
                    PDS_WAVE_LDS_SYNC();
#pragma unroll
                    for (int i = 0; i < SPPC; ++i) put(pa1[i], S[t * SS + i]);
                }
                put(pa0[SPPC], vx0);
                put(pa1[SPPC], vx1);
                put(psj0, vc0);
                put(psj1, vc1);
                put(pnn, nn);
                put(psy, vys);
                put64(pgid, g);
            };
            switch (npend) {
                case 0: into_row(std::integral_constant<int, 0>{}); break;
                case 1: into_row(std::integral_constant<int, 1>{}); break;
                case 2: into_row(std::integral_constant<int, 2>{}); break;
                default: into_row(std::integral_constant<int, 3>{}); break;
            }
            ++npend;
            if (npend == 4) {
                if (debug & 4) npend = 0;  // (timing experiment: routed, never solved)
                else solve_pending();
            }
        }
    };
    // PAIRED, streaming wave: the finished group's moments -> the pair's slot (MidPacked), published under a sequence number.  A slot
    // the solving wave has not taken yet (it is in the middle of a solve) does not stop the stream: the group waits in a spare set of
    // registers and goes out in front of the next one -- the stream stands still only when the solver is two groups behind.
    unsigned pseq = 0;
    d4 st_acc[NPAIR];
    double st_x0 = 0.0, st_x1 = 0.0, st_c0 = 0.0, st_c1 = 0.0, st_nn = 0.0, st_ys = 0.0;
    int64_t st_g = 0;
    bool stashed = false;
    auto slot_free = [&]() __attribute__((always_inline)) {
        return (unsigned)((int)pseq - __builtin_amdgcn_readfirstlane((int)FL[1])) < (unsigned)NSLOT;
    };
    auto publish_group = [&](const d4 (&A)[NPAIR], double vx0, double vx1, double vc0, double vc1, double nn, double vys, int64_t gid)
                             __attribute__((always_inline)) {
        if constexpr (PAIRED) {
            typedef __attribute__((address_space(3))) double* lds_dp;
            using PK = MidPacked<SPPC, YC>;
            const lds_dp S = (lds_dp)(sm + IMG + (DIRECT ? (int)(pseq % NSLOT) * SLOTB : 0));
            if (debug & 16) return;
            PDS_MT(tw);
            while (!slot_free()) __builtin_amdgcn_s_sleep(1);
            PDS_MADD(4, tw);
            PDS_MT(tpb);
            int l0 = fi, l1 = fk, l2 = qb, l3 = qj;
            if constexpr (DIRECT && (SPPC == 32 || ES == 4 || NQ != 1 || PDS_MID_LAUNDER_ALL)) asm volatile("" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));
            const int fi = l0, fk = l1, qb = l2, qj = l3;
            if constexpr (NQ == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = fk + 4 * r;
                    if (i <= fi) S[PK::tri_rt(i) - i + fi] = A[0][r];
                }
                const int par = qb & 1, qr = 4 * qb + fk, c0 = 16 + 4 * par + qj, c1 = 16 + 4 * (par ^ 1) + qj;
                S[PK::tri_rt(qr) - qr + c0] = A[1][0];  // rows 0 .. 15 against the octet: all above the diagonal
                S[PK::tri_rt(qr) - qr + c1] = A[1][1];
                const int rr = 16 + 4 * par + fk;
                if (qb < 2 && rr <= c0) S[PK::tri_rt(rr) - rr + c0] = A[1][2];  // the diagonal 4 x 4 blocks of the corner
                if (qb == 0) S[PK::tri_rt(rr) - rr + c1] = A[1][3];             // rows 16 .. 19 against columns 20 .. 23
            } else if constexpr (NQ != 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = fk + 4 * r;
                    if (i <= fi) S[PK::tri_rt(i) - i + fi] = A[0][r];
                }
                const int qr = 4 * qb + fk;  // quad lanes: D[i = fk][j = qj] of block qb
#pragma unroll
                for (int u = 0; u < NQ; ++u) S[PK::tri_rt(qr) - qr + 16 + 4 * u + qj] = A[1][u];
                {   // the corner: (quad 0, quad 0) [, (quad 0, quad 1), (quad 1, quad 1)]: the upper triangle of it
                    const int ri = 16 + (NQ == 2 && qb >= 2 ? 4 : 0) + fk, cj = 16 + (NQ == 2 && qb >= 1 ? 4 : 0) + qj;
                    if (qb < (NQ == 2 ? 3 : 1) && ri <= cj) S[PK::tri_rt(ri) - ri + cj] = A[2][0];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = fk + 4 * r;  // block (0, 0): G[i][fi], block (0, 1): G[i][16 + fi], block (1, 1): G[16 + i][16 + fi]
                    if (i <= fi) S[PK::tri_rt(i) - i + fi] = A[0][r];
                    if (16 + fi < PK::QC) {
                        S[PK::tri_rt(i) - i + 16 + fi] = A[1][r];
                        if (i <= fi) S[PK::tri_rt(16 + i) - i + fi] = A[2][r];
                    }
                }
            }
            if constexpr (!YC) {
                if (fk == 0) {
                    S[PK::XY + fi] = vx0;
                    S[PK::CS + fi] = vc0;
                    if (16 + fi < SPPC) {
                        S[PK::XY + 16 + fi] = vx1;
                        S[PK::CS + 16 + fi] = vc1;
                    }
                }
                if (lane == 0) {
                    S[PK::TAIL] = nn;
                    S[PK::TAIL + 1] = vys;
                }
            }
            if (lane == 0) S[PK::GID] = __longlong_as_double((long long)gid);
            PDS_WAVE_LDS_SYNC();
            ++pseq;
            FL[0] = pseq;
            PDS_MADD(5, tpb);
        }
    };
    auto flush_stash = [&]() __attribute__((always_inline)) {
        if constexpr (PAIRED) {
            if (stashed) {
                publish_group(st_acc, st_x0, st_x1, st_c0, st_c1, st_nn, st_ys, st_g);
                stashed = false;
            }
        }
    };
    // STASH2 (up to 24 features, where the streaming wave has the registers): TWO spare sets -- a solve of four (17 600 ticks at 17 features)
    // outlasts a group (10 800) and the slot + one set did not always absorb it: the streaming wave stood 9.6 % of its time in front of a
    // full slot.  A two-deep FIFO in registers: set A / set B, `a_old` says which is the older; drained whenever the slot is free.
    constexpr bool STASH2 = PAIRED && YC && SPPC == 24 && !DIRECT;  // (the direct form has the registers for one spare set)
    d4 st_b[STASH2 ? NPAIR : 1];
    int64_t st_gb = 0;
    int nst = 0;
    bool a_old = true;
    auto drain_one = [&]() __attribute__((always_inline)) {  // (the slot is free, nst > 0)
        if constexpr (STASH2) {
            if (a_old) publish_group(st_acc, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, st_g);
            else publish_group(st_b, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, st_gb);
            a_old = !a_old;
            --nst;
        }
    };
    auto drain_free = [&]() __attribute__((always_inline)) {
        if constexpr (STASH2) {
            while (nst > 0 && slot_free()) drain_one();
        }
    };
    auto hand_over2 = [&]() __attribute__((always_inline)) {
        if constexpr (STASH2) {
            drain_free();
            if (nst == 0 && slot_free()) {
                publish_group(acc, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, g);
                return;
            }
            if (nst == 2) drain_one();  // (both sets taken: this one waits for the slot inside publish_group)
            if (nst == 0) {
#pragma unroll
                for (int b = 0; b < NPAIR; ++b) st_acc[b] = acc[b];
                st_g = g;
                a_old = true;
            } else if (a_old) {
#pragma unroll
                for (int b = 0; b < NPAIR; ++b) st_b[b] = acc[b];
                st_gb = g;
            } else {
#pragma unroll
                for (int b = 0; b < NPAIR; ++b) st_acc[b] = acc[b];
                st_g = g;
            }
            ++nst;
        }
    };
    auto hand_over = [&]() __attribute__((always_inline)) {
        if constexpr (STASH2) {
            hand_over2();
        } else if constexpr (DIRECT) {  // (the ring is the queue: the stream stands still only in front of NSLOT unsolved groups)
            if constexpr (YC) publish_group(acc, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, g);
            else publish_group(acc, rows_sum4(xy[0]), rows_sum4(xy[1]), rows_sum4(cs[0]), rows_sum4(cs[1]), (double)rows_in_acc, rows_sum4(ys), g);
        } else if constexpr (PAIRED) {
            double vx0 = xy[0], vx1 = xy[1], vc0 = cs[0], vc1 = cs[1], vys = ys;
            if constexpr (!YC) {
                vx0 = rows_sum4(vx0);
                vx1 = rows_sum4(vx1);
                vc0 = rows_sum4(vc0);
                vc1 = rows_sum4(vc1);
                vys = rows_sum4(vys);
            }
            flush_stash();
            if (slot_free()) {
                publish_group(acc, vx0, vx1, vc0, vc1, (double)rows_in_acc, vys, g);
            } else {
#pragma unroll
                for (int b = 0; b < NPAIR; ++b) st_acc[b] = acc[b];
                st_x0 = vx0; st_x1 = vx1; st_c0 = vc0; st_c1 = vc1;
                st_nn = (double)rows_in_acc; st_ys = vys; st_g = g;
                stashed = true;
            }
        }
    };
    // `whole`: the group's rows all lie in this wave's range.  The walk knows without comparing 64-bit offsets (vector compares: the
    // scalar unit has none): a group that ENDS inside the range (the in-loop call) is whole iff it also started here -- every group but
    // the wave's first does --, the group left open behind the loop is not.
    bool started_here = gs >= W0;
    auto flush = [&](bool whole) __attribute__((always_inline)) {
        if (debug & 2) {  // (timing experiment: no record stores)
            zero_acc();
            rows_in_acc = 0;
            return;
        }
        if constexpr (SPPC > 0) {
            if (whole) {
                if constexpr (PAIRED) hand_over();
                else route_pending();
            } else {
                // the wave the group starts in owns the side-table slot (largest w whose first row is <= gs)
                int64_t slot = wave;
                if (!started_here) {
                    int64_t lo = 0, hi = wave;
                    while (hi - lo > 1) {
                        const int64_t mid = lo + ((hi - lo) >> 1);
                        const int64_t hm = H0 + (H1 - H0) * mid / nwaves;
                        const int64_t wm = hm * HR > row_begin ? hm * HR : row_begin;
                        if (wm <= gs) lo = mid;
                        else hi = mid;
                    }
                    slot = lo;
                } else if (lane == 0) {
                    sa.side_list[wave] = (int32_t)g;
                }
                put_record(sa.side_rec + slot * (int64_t)q * q, false);
            }
        } else {
            put_record(records + g * (int64_t)q * q, whole);
        }
        zero_acc();
        rows_in_acc = 0;
    };
    // ---- stream the half-tiles
    // OWNER: a group belongs to the wave in whose rows it STARTS.  That wave follows it beyond the end of
    // its own range -- at most `own_limit` rows, less than the next wave's whole range --, the wave it runs into skips it: no partial sums, no
    // atomics, no side record, the same bits every run.  Only a group LONGER than `own_limit` (both waves judge by its offsets alone) is
    // summed by every wave that meets it, into the side table's slot of the wave it starts in, as before.
    constexpr bool OWNER = PDS_MID_OWNER;  // (the record form too: whole records by plain stores, the caller zeroes only giant and empty groups' records)
    const int64_t own_limit = ((H1 - H0) / nwaves) * HR;
    bool skip = OWNER && !started_here && ge - gs <= own_limit;  // the wave's first group, begun (and finished) by the wave in front
    int64_t wend = W1, hend = h1;                                 // the rows this wave walks, the half-tiles it streams: grow with a followed group
    int64_t pos = W0;
    if constexpr (DIRECT) issue_direct(h0);
    else fetch_tile(0, h0);
    for (int64_t h = h0; h < hend; ++h) {
        const int buf = (int)((h - h0) & 1);
        PDS_MT(p0);
        if constexpr (DIRECT) {  // half-tile h has landed: it becomes the set the walk reads (the compiler's own wait sits in front of the moves)
            if constexpr (!YC) {  // (and the target's image, which the compiler does not follow into LDS)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PDS_WAVE_LDS_SYNC();
            }
#pragma unroll
            for (int k = 0; k < NB8; ++k)
#pragma unroll
                for (int b = 0; b < NOP; ++b) {
                    CUR[k][b] = NXT[k][b];
                    // (a definition the compiler cannot see through: without it the first half-tile is loaded into CUR directly, CUR counts as
                    // "maybe in flight" in the whole loop, and every step of the walk waits for the NEXT half-tile's loads)
                    asm volatile("" : "+v"(CUR[k][b]));
                }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // half-tile h has landed (and this wave's record stores are out)
            PDS_WAVE_LDS_SYNC();
        }
        PDS_MADD(0, p0);
        // (a waiting group goes out as soon as the solving wave has emptied the slot, not only when the next group ends: without this look
        // per half-tile the 17-feature kernel is 6 % slower)
        PDS_MT(p11);
        if constexpr (STASH2) {
            if (nst > 0) drain_free();
        } else if constexpr (PAIRED && !DIRECT) {
            if (stashed && slot_free()) flush_stash();
        }
        PDS_MADD(11, p11);
        PDS_MT(p1);
        if (h + 1 < (OWNER ? H1 : h1)) {  // (OWNER: also the half-tile behind the wave's own range -- a group may have to be followed into it)
            if constexpr (DIRECT) issue_direct(h + 1);
            else fetch_tile(buf ^ 1, h + 1);  // (the other image was consumed one iteration ago)
        }
        PDS_MADD(1, p1);
        const int64_t R0 = h * HR;
        const int64_t tile_end = R0 + HR < wend ? R0 + HR : wend;
        // the walk inside a half-tile runs on 32-bit offsets relative to its first row (scalar compares; as int64 every `pos < tile_end`,
        // `ge < tile_end`, `pos == ge` was a vector compare + a branch on its result): the open group's end is clamped to HR + 1
        auto rel_end = [&](int64_t e) __attribute__((always_inline)) {  // e > R0
            const uint64_t d = (uint64_t)(e - R0);
            const uint32_t dh = (uint32_t)(d >> 32), dl = (uint32_t)d;
            return (dh != 0u || dl > (uint32_t)HR) ? HR + 1 : (int)dl;
        };
        const int tile_n = (int)(tile_end - R0);
        int posr = (int)(pos - R0);
        int ger = rel_end(ge);
        while (posr < tile_n) {
            const int seg = ger < tile_n ? ger : tile_n;
            PDS_MT(p2);
            if (seg > posr && !(debug & 1) && !skip) consume(buf, posr, seg);
            PDS_MADD(2, p2);
            if (debug & 1) rows_in_acc += seg - posr;
            posr = seg;
            if (posr == ger) {  // group complete (as far as this wave's rows go: `started_here` decides how it is written)
                PDS_MT(p3);
                if (rows_in_acc != 0) flush(started_here);
                PDS_MADD(3, p3);
                PDS_MT(p6);
                do {  // (empty groups: their records stay zero)
                    ++g;
                    gs = ge;
                    ge = ge_next;
                    ge_next = off[g + 2 <= n_groups ? g + 2 : n_groups];
                } while (g < n_groups && ge == gs);
                started_here = true;
                skip = false;
                PDS_MADD(6, p6);
                if (g >= n_groups) break;
                ger = rel_end(ge);
            }
        }
        pos = R0 + posr;
        if (g >= n_groups) break;
        if constexpr (OWNER) {
            // the open group at the end of the wave's (possibly extended) rows: its own and not longer than the limit -> one more half-tile
            if (h + 1 == hend && rows_in_acc != 0 && started_here && ge - gs <= own_limit && h + 1 < H1) {
                wend = ge < row_end ? ge : row_end;
                hend = h + 2;
            }
        }
        PDS_WAVE_LDS_SYNC();
    }
    if (rows_in_acc != 0 && g < n_groups) flush(false);  // the group that continues in the next wave's rows
    if constexpr (PAIRED) {
        flush_stash();
        if constexpr (STASH2)
            while (nst > 0) drain_one();  // (publish_group waits for the slot)
        while ((pseq & 3u) != 0u && !(debug & 16)) {  // pad the last batch: the slot's contents once more (a valid system), marked as discarded
            typedef __attribute__((address_space(3))) double* lds_dp;
            while (!slot_free()) __builtin_amdgcn_s_sleep(1);
            if (lane == 0) ((lds_dp)(sm + IMG + (DIRECT ? (int)(pseq % NSLOT) * SLOTB : 0)))[MidPacked<SPPC, YC>::GID] = __longlong_as_double(-1ll);
            PDS_WAVE_LDS_SYNC();
            ++pseq;
            FL[0] = pseq;
        }
        PDS_WAVE_LDS_SYNC();
        FL[2] = 1u;
    } else {
        solve_pending();
    }
#ifdef PDS_PROFILE_MID
    PDS_MADD(7, t_kernel);
    mprof_out();
#endif
#undef PDS_GM_LDSD
#undef PDS_GM_LDST
}

// the records the stream ADDS to (a group longer than a wave's range: summed by every wave it meets) or never writes (groups without rows):
// zeroed here, sixty-four groups per block; every other record is written once, whole, by plain stores -- the memset of all records this
// replaces was 3.5 GB per call at 100 000 groups x 64 features.  `limit` is the kernel's own_limit, from the same numbers.
__global__ __launch_bounds__(256) void mid_zero_records_kernel(const int64_t* __restrict__ off, int64_t n_groups, int qq, double* __restrict__ records,
                                                               int hr, int64_t nwaves) {
    const int64_t row_begin = off[0], row_end = off[n_groups];
    const int64_t H0 = row_begin / hr, H1 = (row_end + hr - 1) / hr;
    const int64_t limit = row_end > row_begin ? ((H1 - H0) / nwaves) * hr : -1;
    for (int k = 0; k < 64; ++k) {
        const int64_t g = (int64_t)blockIdx.x * 64 + k;
        if (g >= n_groups) return;
        const int64_t len = off[g + 1] - off[g];
        if (len != 0 && len <= limit) continue;
        for (int e = threadIdx.x; e < qq; e += 256) records[g * (int64_t)qq + e] = 0.0;
    }
}
template <int NBLK>
int launch_grouped_stream(pds_ctx* ctx, const DeviceCols<double>& dc, int p, int64_t n_frame, const int64_t* d_off, int64_t n_groups,
                          double* d_records) {
    using MD = MidDims<NBLK>;
    const int q = p + 2;
    // 33 .. 64 features: the direct form too (round 6) -- a wave alone on its SIMD has the registers for two operand sets of four pieces
    // beside ten accumulator blocks, and no tile images at all (the target's image: 1 KB)
    constexpr bool kDirect = NBLK == 4 && PDS_MID_DIRECT_RECORDS;
    constexpr int hr = kDirect ? kMidDirectRows<8> : MD::HR, lds_bytes = kDirect ? 2 * kMidDirectYBytes + 64 : MD::LDS_BYTES;
    auto kern = grouped_mid_stream_kernel<NBLK, 0, false, false, 0, double, kDirect>;
    if (lds_bytes > 64 * 1024)
        PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    // a wave takes at least eight half-tiles: a group then meets at most two waves unless it is longer than a wave's whole range, and
    // the sum of two partial records does not depend on their order -- results are reproducible run to run but for such giant groups
    // (the row count of the chunk is not known on the host: the frame's is an upper bound)
    const int64_t waves = std::min<int64_t>((int64_t)ctx->num_cus * kMidWavesPerCu, std::max<int64_t>(1, n_frame / (8 * hr)));
    if (PDS_MID_OWNER)
        hipLaunchKernelGGL(mid_zero_records_kernel, dim3((unsigned)((n_groups + 63) / 64)), dim3(256), 0, ctx->stream, d_off, n_groups, q * q, d_records, hr,
                           waves);
    else
        PDS_HIP_CHECK(hipMemsetAsync(d_records, 0, (size_t)n_groups * q * q * sizeof(double), ctx->stream));
#ifdef PDS_DEV_SWITCHES  // timing experiments of development builds (EXTRA=-DPDS_DEV_SWITCHES): wrong results with it
    const char* dbg = dev_env("PDS_GMID_DEBUG");
#else
    const char* dbg = nullptr;
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)waves), dim3(64), lds_bytes, ctx->stream, dc.d_ptrs, p, n_frame, d_off, n_groups, d_records,
                       dbg ? std::atoi(dbg) : 0, MidSolveArgs{});
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

// ---- the in-wave-solve form (SPPC): host side
constexpr unsigned kMidMarkCap = 8192;  // records of marked systems kept for the pivoted QR; more than that: the record pipeline
inline int64_t mid_fused_waves(const pds_ctx* ctx, int64_t n_frame, int hr = MidDims<2>::HR) {
    const int64_t w = std::min<int64_t>((int64_t)ctx->num_cus * kMidWavesPerCu, std::max<int64_t>(1, n_frame / (8 * hr)));
    // (whole workgroups of four pairs: the PAIRED form -- also for frames of fewer than four waves' worth of rows, where some pairs find no
    // half-tile and leave at once: small f32 frames used to fall through to f32 moment RECORDS, whose 6e-8 moments cannot see a pivot
    // of 1e-11 under a diagonal of 300 -- tools/experiments/f32_gate_case.py)
    return w >= 4 ? w / 4 * 4 : 4;
}
// side table -> compact: the groups that straddle wave boundaries (slot w used iff side_list[w] >= 0), their records, row counts as
// offsets, and the count (there are at most `waves` <= 1024 of them: every block repeats the scan, block 0 writes the lists, all blocks
// share the copy of the records -- one block alone took 0.5 ms over the 9.5 MB of 1 024 records at 32 features)
__global__ __launch_bounds__(1024) void mid_side_compact_kernel(const double* __restrict__ side_rec, const int32_t* __restrict__ side_list,
                                                                int waves, int qq, const int64_t* __restrict__ off, double* __restrict__ rec_c,
                                                                int32_t* __restrict__ list_c, int64_t* __restrict__ rows_c,
                                                                unsigned* __restrict__ count_out) {
    // thread w = slot w (waves <= 1024): inclusive scans of "used" and of the row counts through shared memory
    __shared__ int s_cnt[1024];
    __shared__ long long s_rows[1024];
    __shared__ int s_slot[1024];
    const int w = threadIdx.x;
    const int32_t g = w < waves ? side_list[w] : -1;
    const long long rows = g >= 0 ? (long long)(off[g + 1] - off[g]) : 0;
    s_cnt[w] = g >= 0 ? 1 : 0;
    s_rows[w] = rows;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int c = w >= d ? s_cnt[w - d] : 0;
        const long long r = w >= d ? s_rows[w - d] : 0;
        __syncthreads();
        s_cnt[w] += c;
        s_rows[w] += r;
        __syncthreads();
    }
    const int n = s_cnt[1023];
    if (g >= 0) {
        const int k = s_cnt[w] - 1;
        s_slot[k] = w;
        if (blockIdx.x == 0) {
            list_c[k] = g;
            rows_c[k + 1] = s_rows[w];
        }
    }
    if (w == 0 && blockIdx.x == 0) {
        rows_c[0] = 0;
        *count_out = (unsigned)n;
    }
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * 1024 + w; e < (int64_t)n * qq; e += (int64_t)gridDim.x * 1024) {
        const int k = (int)(e / qq);
        rec_c[e] = side_rec[(int64_t)s_slot[k] * qq + (e - (int64_t)k * qq)];
    }
}
// groups without rows: nobody finishes them, so nobody answers them -- null, NaN coefficients (the fill of the whole coefficient block
// this replaces was 136 .. 264 MB of stores per call at 1e6 groups)
template <typename T>
__global__ __launch_bounds__(256) void mid_empty_groups_kernel(const int64_t* __restrict__ off, int64_t n_groups, int pp, T* __restrict__ coeffs,
                                                               uint8_t* __restrict__ flags) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_groups || off[g + 1] != off[g]) return;
    flags[g] = 1;
    for (int c = 0; c < pp; ++c) coeffs[g * pp + c] = (T)__builtin_nan("");
}
template <typename T>
__global__ __launch_bounds__(256) void mid_scatter_kernel(const double* __restrict__ co_c, const uint8_t* __restrict__ fl_c,
                                                          const int32_t* __restrict__ list, int64_t n, int pp, T* __restrict__ coeffs,
                                                          uint8_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * pp) return;
    const int64_t k = i / pp;
    const int c = (int)(i - k * pp);
    const int64_t g = list[k];
    coeffs[g * pp + c] = (T)co_c[i];
    if (c == 0) flags[g] = fl_c[k] ? 1 : 0;
}

}  // namespace

// Moment records ((p+2)^2 doubles each, column-major over [x_0 .. x_{p-1}, 1, y]) of n_groups contiguous groups (d_off: n_groups + 1
// device offsets into the frame of n_frame rows) with 17 .. 64 f64 features, from one stream over the groups' rows.
int launch_grouped_moments_stream(pds_ctx* ctx, const DeviceCols<double>& dc, int n_feat, int64_t n_frame, const int64_t* d_off, int64_t n_groups,
                                  double* d_records) {
    if (n_groups <= 0) return PDS_OK;
    KernelTimer timer(ctx, kKindGroupedMoments);
    if (n_feat <= 32) return launch_grouped_stream<2>(ctx, dc, n_feat, n_frame, d_off, n_groups, d_records);
    if (n_feat <= 64) return launch_grouped_stream<4>(ctx, dc, n_feat, n_frame, d_off, n_groups, d_records);
    return fail(PDS_ERR_UNSUPPORTED, "grouped_mid_stream: up to 64 features");
}

// OLS / ridge fits of n_groups contiguous groups with 17 .. 32 f64 features, rank gate on, as ONE stream with the solves in the
// streaming waves (grouped_mid_stream_kernel, SPPC): coefficients [n_groups][p + bias] and null flags; no per-group records.
// PDS_ERR_UNSUPPORTED (nothing usable written): not applicable, or more systems next to the gate than the marked list holds -- the
// caller keeps the record pipeline.  d_ws: grouped_mid_fused_workspace() bytes.
size_t grouped_mid_fused_workspace(int num_cus, int n_feat, int add_bias) {
    const size_t q = (size_t)n_feat + 2, pp = (size_t)n_feat + (add_bias ? 1 : 0), waves = (size_t)num_cus * kMidWavesPerCu;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t sysmax = std::max<size_t>(waves, kMidMarkCap);
    return 4096 + 2 * up(waves * q * q * 8) + 2 * up(waves * 4) + up((waves + 1) * 8) + up((size_t)kMidMarkCap * q * q * 8) + up((size_t)kMidMarkCap * 4) +
           up(sysmax * pp * 8) + up(sysmax) + solve_wave_workspace(n_feat, add_bias, (int64_t)waves, 8) + 512;
}
template <typename T>
int launch_grouped_mid_fused(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_frame, const int64_t* d_off, int64_t n_groups,
                             const SolveParams& sp, T* d_coeffs, uint8_t* d_flags, void* d_ws) {
    if (n_feat <= 16 || n_feat > 32 || !(sp.gate_tol > 0.0) || sp.lambda_on_bias || !d_flags || !d_ws || n_groups <= 0 ||
        n_groups >= (1ll << 31))
        return PDS_ERR_UNSUPPORTED;
    using MD = MidDims<2, (int)sizeof(T)>;
    constexpr bool F64 = sizeof(T) == 8;
    const int p = n_feat, q = p + 2, bias = sp.add_bias ? 1 : 0, pp = p + bias;
    const int64_t waves = mid_fused_waves(ctx, n_frame, MD::HR);
    if (waves > 1024) return PDS_ERR_UNSUPPORTED;  // (mid_side_compact_kernel: one thread per wave)
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    char* wsp = static_cast<char*>(d_ws);
    wsp += (256 - (reinterpret_cast<uintptr_t>(wsp) & 255)) & 255;
    auto take = [&](size_t b) { char* r = wsp; wsp += up(b); return r; };
    MidSolveArgsT<T> sa;
    sa.sp.p = p;
    sa.sp.pp = p;
    sa.sp.bias = bias;
    sa.sp.lambda_on_bias = 0;
    sa.sp.lambda = sp.lambda;
    sa.sp.gate_on = 1;
    sa.sp.ln_tol = std::log(sp.gate_tol);
    sa.sp.inv_tol = 1.0 / sp.gate_tol;
    const bool second_pass = sp.solver != PDS_SOLVER_CHOLESKEY;  // (as launch_solve_wave: "choleskey" IS the in-wave factorisation)
    sa.sp.sus_tol = second_pass ? std::sqrt(sa.sp.inv_tol) : 0.0;
    sa.sp.sus_ratio = second_pass ? solve_suspect_ratio() : 0.0;
    sa.sp.sus_band = 1e-5;
    sa.coeffs = d_coeffs;
    sa.flags = d_flags;
    unsigned* d_counts = reinterpret_cast<unsigned*>(take(256));  // [0] marked, [1] side groups
    sa.mark_count = d_counts;
    sa.side_rec = reinterpret_cast<double*>(take((size_t)waves * q * q * 8));
    sa.side_list = reinterpret_cast<int32_t*>(take((size_t)waves * 4));
    sa.mark_rec = reinterpret_cast<double*>(take((size_t)kMidMarkCap * q * q * 8));
    sa.mark_list = reinterpret_cast<int32_t*>(take((size_t)kMidMarkCap * 4));
    sa.mark_cap = kMidMarkCap;
    double* rec_c = reinterpret_cast<double*>(take((size_t)waves * q * q * 8));
    int32_t* list_c = reinterpret_cast<int32_t*>(take((size_t)waves * 4));
    int64_t* rows_c = reinterpret_cast<int64_t*>(take((size_t)(waves + 1) * 8));
    const size_t sysmax = std::max<size_t>((size_t)waves, kMidMarkCap);
    double* co_c = reinterpret_cast<double*>(take(sysmax * pp * 8));
    uint8_t* fl_c = reinterpret_cast<uint8_t*>(take(sysmax));
    void* wave_ws = take(solve_wave_workspace(n_feat, bias, waves, 8));
    // groups nobody answers (no rows) are null with NaN coefficients: the kernel writes a group's answer where it finishes it
    hipLaunchKernelGGL(mid_empty_groups_kernel<T>, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, ctx->stream, d_off, n_groups, pp, d_coeffs,
                       d_flags);
    PDS_HIP_CHECK(hipMemsetAsync(d_counts, 0, 256, ctx->stream));
    PDS_HIP_CHECK(hipMemsetAsync(sa.side_rec, 0, (size_t)waves * q * q * 8, ctx->stream));
    PDS_HIP_CHECK(hipMemsetAsync(sa.side_list, 0xFF, (size_t)waves * 4, ctx->stream));
    constexpr int lds = MD::LDS_BYTES + kMidSolveScratch;
    const char* pair_env = dev_env("PDS_GROUPED_MID_PAIRED");
    const bool paired = !(pair_env && pair_env[0] == '0') && waves % 4 == 0;
    if (!F64 && !paired) return PDS_ERR_UNSUPPORTED;  // (f32 frames: the paired form only)
    {
        KernelTimer timer(ctx, kKindGroupedMoments);
#ifdef PDS_DEV_SWITCHES  // timing experiments of development builds (EXTRA=-DPDS_DEV_SWITCHES): wrong results with it
        const char* dbg = dev_env("PDS_GMID_DEBUG");
#else
        const char* dbg = nullptr;
#endif
        const int debug = dbg ? std::atoi(dbg) : 0;
        auto launch = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)waves), dim3(64), lds, ctx->stream, dc.d_ptrs, p, n_frame, d_off, n_groups, (double*)nullptr, debug, sa);
        };
        bool paired_lds_ok = true;
        auto launch_paired_c = [&](auto kern, auto direct_c) {
            constexpr int plds = 4 * kMidPairLds<2, decltype(direct_c)::value>;
            // (per call, not once per process: the attribute belongs to the function ON THE CURRENT DEVICE, and a process may drive several)
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, plds) != hipSuccess) {
                (void)hipGetLastError();
                paired_lds_ok = false;
                return;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)(waves / 4)), dim3(512), plds, ctx->stream, dc.d_ptrs, p, n_frame, d_off, n_groups, (double*)nullptr,
                               debug, sa);
        };
        auto launch_paired = [&](auto kern) { launch_paired_c(kern, std::false_type{}); };
        if (paired) {
            const char* yc_env = dev_env("PDS_GROUPED_MID_YC");  // (development: '0' keeps the side sums of the 31 / 32-feature form)
            const bool yc = !F64 || !(yc_env && yc_env[0] == '0');  // (f32 frames have the ones / target column form only)
            const char* nq_env = dev_env("PDS_GROUPED_MID_QUAD");  // (development: '0' keeps the 16 x 16 x 4 form of the second block)
            // the direct form (f64 frames, up to 30 features: PDS_MID_DIRECT, on by default; PDS_GROUPED_MID_DIRECT=0 in development builds)
            const char* dir_env = dev_env("PDS_GROUPED_MID_DIRECT");
            const bool direct = PDS_MID_DIRECT && yc && (F64 || p <= 30) && !(dir_env && dir_env[0] == '0');  // (f32 frames of 31 / 32 features: the LDS form)
            if (direct) {
                if (p <= 18) launch_paired_c(grouped_mid_stream_kernel<2, 24, true, true, 1, T, true>, std::true_type{});
                else if (p <= 22 && PDS_MID_DIRECT_OCTET) launch_paired_c(grouped_mid_stream_kernel<2, 24, true, true, 3, T, true>, std::true_type{});
                else if (p <= 24) launch_paired_c(grouped_mid_stream_kernel<2, 24, true, true, 0, T, true>, std::true_type{});
                else if (p <= 30) launch_paired_c(grouped_mid_stream_kernel<2, 32, true, true, 0, T, true>, std::true_type{});
                else if constexpr (F64) launch_paired_c(grouped_mid_stream_kernel<2, 32, true, false, 0, T, true>, std::true_type{});
            }
            if (direct) {
            } else
            if (p <= 18 && yc && !(nq_env && nq_env[0] == '0')) launch_paired(grouped_mid_stream_kernel<2, 24, true, true, 1, T>);
            else if (p <= 22 && yc && !(nq_env && nq_env[0] == '0')) launch_paired(grouped_mid_stream_kernel<2, 24, true, true, 2, T>);
            else if (p <= 24 && (yc || !F64)) launch_paired(grouped_mid_stream_kernel<2, 24, true, true, 0, T>);
            else if (p <= 24) {
                if constexpr (F64) launch_paired(grouped_mid_stream_kernel<2, 24, true, false>);
            } else if (p <= 30 && yc) launch_paired(grouped_mid_stream_kernel<2, 32, true, true, 0, T>);
            else launch_paired(grouped_mid_stream_kernel<2, 32, true, false, 0, T>);
        } else if constexpr (F64) {
            if (p <= 24) launch(grouped_mid_stream_kernel<2, 24>);
            else launch(grouped_mid_stream_kernel<2, 32>);
        }
        if (!paired_lds_ok) return PDS_ERR_UNSUPPORTED;  // (a device that does not grant a workgroup the whole LDS: the record pipeline)
        PDS_HIP_CHECK(hipGetLastError());
    }
    // the groups cut by wave boundaries: compacted, then the record solver
    hipLaunchKernelGGL(mid_side_compact_kernel, dim3(64), dim3(1024), 0, ctx->stream, (const double*)sa.side_rec, (const int32_t*)sa.side_list,
                       (int)waves, q * q, d_off, rec_c, list_c, rows_c, d_counts + 1);
    unsigned h_counts[2] = {0, 0};
    PDS_HIP_CHECK(hipMemcpyAsync(h_counts, d_counts, sizeof(h_counts), hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
#ifdef PDS_DEV_SWITCHES
    if (dev_env("PDS_GMID_VERBOSE")) std::fprintf(stderr, "grouped_mid_fused: waves %lld marked %u side %u\n", (long long)waves, h_counts[0], h_counts[1]);
#endif
    if (h_counts[0] > kMidMarkCap) return PDS_ERR_UNSUPPORTED;  // (every group will be answered by the record pipeline instead)
    if (h_counts[1] > 0) {
        const int64_t ns = h_counts[1];
        if (int rc = launch_solve_wave<double>(ctx, rec_c, ns, sp, co_c, fl_c, rows_c, wave_ws)) return rc;
        hipLaunchKernelGGL(mid_scatter_kernel<T>, dim3((unsigned)((ns * pp + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)co_c,
                           (const uint8_t*)fl_c, (const int32_t*)list_c, ns, pp, d_coeffs, d_flags);
    }
    if (h_counts[0] > 0) {  // systems next to the gate: the reference's default factorisation (pivoted QR, log-det gate)
        const int64_t nm = h_counts[0];
        if (int rc = launch_solve_marked<double>(ctx, sa.mark_rec, nm, sp, co_c, fl_c, nullptr)) return rc;
        hipLaunchKernelGGL(mid_scatter_kernel<T>, dim3((unsigned)((nm * pp + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)co_c,
                           (const uint8_t*)fl_c, (const int32_t*)sa.mark_list, nm, pp, d_coeffs, d_flags);
    }
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
template int launch_grouped_mid_fused<double>(pds_ctx*, const DeviceCols<double>&, int, int64_t, const int64_t*, int64_t, const SolveParams&, double*, uint8_t*,
                                              void*);
template int launch_grouped_mid_fused<float>(pds_ctx*, const DeviceCols<float>&, int, int64_t, const int64_t*, int64_t, const SolveParams&, float*, uint8_t*,
                                             void*);

#ifdef PDS_PROFILE_MID
extern "C" int pds_debug_mid_phase_cycles(unsigned long long* out, int reset) {
    static const unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mid_phase), sizeof(z)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_mid_phase), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

}  // namespace pds
