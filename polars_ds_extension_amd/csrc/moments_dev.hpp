// moments_dev.hpp -- device-side building blocks of the Gram build, shared by moments.hip (single system,
// grouped) and grouped_fused.hip (Gram + solve in one kernel).  See moments.hip for the design notes.
#pragma once
#include "common.hpp"

namespace pds {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kWaves = 4;           // waves per block; each wave is independent (no __syncthreads in the loop)
constexpr int kColStride = 1040;    // bytes per LDS tile column (1 KiB + 16 B pad)
constexpr int kSlots = 18;          // 16 features, y, w
constexpr int kSlotY = 16, kSlotW = 17;
constexpr int kWaveLds = kSlots * kColStride;  // 18720 B

template <typename T>
struct Tile;
template <>
struct Tile<double> {
    using vec = d2u;
    using acc = d4;
    static constexpr int RPL = 2;  // rows per lane per 16-byte load
    static __device__ __forceinline__ acc mfma(double a, double b, acc c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    static __device__ __forceinline__ int drow(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <>
struct Tile<float> {
    using vec = f4u;
    using acc = f4;
    static constexpr int RPL = 4;
    static __device__ __forceinline__ acc mfma(float a, float b, acc c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4) * 4 + reg
    static __device__ __forceinline__ int drow(int lane, int reg) { return (lane >> 4) * 4 + reg; }
};

// Per-lane accumulators of one wave.  D is kept in double for both precisions: the f32 path
// accumulates one tile (256 rows) in f32 on the matrix core and folds it into these per tile.
struct WaveAcc {
    double d[4];
    double xy, cs, yy, ys, sw;
};

template <typename T>
struct TileRegs {
    typename Tile<T>::vec x[16];
    typename Tile<T>::vec y, w;
};

// The column base pointers are loop invariant and wave uniform: fetch them ONCE into SGPRs.  (Left to the
// compiler, every tile re-issued 17 dependent s_load + s_waitcnt pairs in front of the global loads.)
template <typename T>
struct ColPtrs {  // global address space: see gptr in common.hpp
    gptr<T> x[16];
    gptr<T> y;
    gptr<T> w;
};
template <typename T, bool WEIGHTED>
__device__ __forceinline__ void fetch_col_ptrs(const T* const* __restrict__ cols, int p, ColPtrs<T>& cp) {
#pragma unroll
    for (int c = 0; c < 16; ++c) cp.x[c] = as_global(cols[c < p ? c : 0]);
    cp.y = as_global(cols[p]);
    cp.w = as_global(WEIGHTED ? cols[p + 1] : cols[p]);
}

// Streaming loads: every input element is read exactly once, so the loads are issued non-temporal (`nt`: no allocation
// in L2 / MALL for lines nobody will ask for again).  Measured with the kernels' own access pattern and no math
// (tools/membw.hip, 17 column streams of 1e8 f64 rows, wave-owned row ranges): 6.09 TB/s plain, 6.78 TB/s non-temporal.
// -DPDS_NO_NT builds the plain loads (A/B).
#ifdef PDS_NO_NT
#define PDS_STREAM_LOAD(q) (*(q))
#else
#define PDS_STREAM_LOAD(q) __builtin_nontemporal_load(q)
#endif
// (a macro, not a template: deducing the vector type drops the under-alignment attribute of d2u / f4u -- column pointers are
//  only element aligned)

// ---- full-tile load: lane reads RPL consecutive rows of every column (16 B, coalesced 1 KiB/instr)
template <typename T, bool WEIGHTED>
__device__ __forceinline__ void load_full_tile(const ColPtrs<T>& cp, int p, int64_t row, TileRegs<T>& r) {
    using V = typename Tile<T>::vec;
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (c < p) r.x[c] = PDS_STREAM_LOAD(reinterpret_cast<gptr<V>>(cp.x[c] + row));
    r.y = PDS_STREAM_LOAD(reinterpret_cast<gptr<V>>(cp.y + row));
    if (WEIGHTED) r.w = PDS_STREAM_LOAD(reinterpret_cast<gptr<V>>(cp.w + row));
}

// ---- guarded load for the ragged last tile (rows >= n contribute exact zeros)
template <typename T, bool WEIGHTED>
__device__ __forceinline__ void load_tail_tile(const ColPtrs<T>& cp, int p, int64_t row, int64_t n,
                                               TileRegs<T>& r) {
    constexpr int RPL = Tile<T>::RPL;
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (c < p) {
#pragma unroll
            for (int e = 0; e < RPL; ++e) r.x[c][e] = (row + e < n) ? cp.x[c][row + e] : T(0);
        }
#pragma unroll
    for (int e = 0; e < RPL; ++e) r.y[e] = (row + e < n) ? cp.y[row + e] : T(0);
    if (WEIGHTED) {
#pragma unroll
        for (int e = 0; e < RPL; ++e) r.w[e] = (row + e < n) ? cp.w[row + e] : T(0);
    }
}

template <typename T, bool WEIGHTED>
__device__ __forceinline__ void store_tile_lds(char* wl, int p, int lane, const TileRegs<T>& r) {
    using V = typename Tile<T>::vec;
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (c < p) *reinterpret_cast<V*>(wl + c * kColStride + lane * 16) = r.x[c];
    *reinterpret_cast<V*>(wl + kSlotY * kColStride + lane * 16) = r.y;
    if (WEIGHTED) *reinterpret_cast<V*>(wl + kSlotW * kColStride + lane * 16) = r.w;
}

// ---- consume `steps` groups of 4 rows from the wave's LDS tile
// LEAN (weighted builds whose record is only read as a MEAT block: the HC passes of lin_reg_report): 1 = X'y and y'y are not formed,
// 2 = sum w and sum w y are not formed either (the caller adds them once per TILE from the lane's own rows) -- three / five of the
// eight f64 vector operations per step.  They are not free beside the matrix instructions: a v_fma_f64 takes the SIMD's FP64 unit
// for 8.6 clk, a v_mfma_f64_16x16x4 for 65, and the two take turns (tools/fp64_share_probe.hip, profiles/r06_fp64_share_probe.txt).
template <typename T, bool WEIGHTED, int LEAN = 0>
__device__ __forceinline__ void consume_tile(const char* wl, int lane, int steps, WaveAcc& a) {
    const int f = lane & 15, q = lane >> 4;
    const T* xcol = reinterpret_cast<const T*>(wl + f * kColStride) + q;
    const T* ycol = reinterpret_cast<const T*>(wl + kSlotY * kColStride) + q;
    const T* wcol = reinterpret_cast<const T*>(wl + kSlotW * kColStride) + q;
    if constexpr (sizeof(T) == 8) {
        d4 acc = {a.d[0], a.d[1], a.d[2], a.d[3]};
        double xy = a.xy, cs = a.cs, yy = a.yy, ys = a.ys, sw = a.sw;
        // software pipeline: the operands of step s+2 are fetched from LDS before the MFMA of step s issues, so
        // the ds_read latency sits under two matrix instructions instead of in front of each one (reading two
        // steps past the end stays inside the wave's LDS allocation and is discarded)
        double xn0 = xcol[0], yn0 = ycol[0], wn0 = WEIGHTED ? wcol[0] : 0.0;
        double xn1 = xcol[4], yn1 = ycol[4], wn1 = WEIGHTED ? wcol[4] : 0.0;
#pragma unroll 8
        for (int s = 0; s < steps; ++s) {
            const double x = xn0, yv = yn0, wv0 = wn0;
            xn0 = xn1; yn0 = yn1; wn0 = wn1;
            xn1 = xcol[4 * s + 8];
            yn1 = ycol[4 * s + 8];
            if (WEIGHTED) wn1 = wcol[4 * s + 8];
            double xa = x;
            if (WEIGHTED) {
                double wv = wv0;
                xa = x * wv;
                if (LEAN == 0) yy = fma(wv * yv, yv, yy);
                if (LEAN < 2) {
                    ys = fma(wv, yv, ys);
                    sw += wv;
                }
            } else {
                yy = fma(yv, yv, yy);
                ys += yv;
            }
            acc = Tile<double>::mfma(xa, x, acc);
            if (LEAN == 0) xy = fma(xa, yv, xy);
            cs += xa;
        }
        a.d[0] = acc[0]; a.d[1] = acc[1]; a.d[2] = acc[2]; a.d[3] = acc[3];
        a.xy = xy; a.cs = cs; a.yy = yy; a.ys = ys; a.sw = sw;
    } else {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        float xy = 0.f, cs = 0.f, yy = 0.f, ys = 0.f, sw = 0.f;
#pragma unroll 8
        for (int s = 0; s < steps; ++s) {
            float x = xcol[4 * s];
            float yv = ycol[4 * s];
            float xa = x;
            if (WEIGHTED) {
                float wv = wcol[4 * s];
                xa = x * wv;
                if (LEAN == 0) yy = fmaf(wv * yv, yv, yy);
                if (LEAN < 2) {
                    ys = fmaf(wv, yv, ys);
                    sw += wv;
                }
            } else {
                yy = fmaf(yv, yv, yy);
                ys += yv;
            }
            acc = Tile<float>::mfma(xa, x, acc);
            if (LEAN == 0) xy = fmaf(xa, yv, xy);
            cs += xa;
        }
        a.d[0] += (double)acc[0]; a.d[1] += (double)acc[1]; a.d[2] += (double)acc[2]; a.d[3] += (double)acc[3];
        a.xy += (double)xy; a.cs += (double)cs; a.yy += (double)yy; a.ys += (double)ys; a.sw += (double)sw;
    }
}


// p <= 8 features: SEVERAL 4-row slabs per matrix instruction.  With P2 = 8 / 4 / 2 / 1 feature slots in use, operand
// column f carries feature f % P2 of row slab f / P2, so one v_mfma 16x16x4 consumes 4 * (16 / P2) rows; the diagonal
// P2 x P2 blocks of the product are the slabs' Gram contributions (the off-diagonal blocks are cross terms nobody reads)
// and are folded by the finalize kernel.  At p <= 8 the f64 matrix pipe (86 clk per instruction) is the bound of the
// unpacked kernel, not HBM: 1e8 rows cost 1.2 ms whether p is 1 or 8.
template <typename T, bool WEIGHTED, int P2>
__device__ __forceinline__ void consume_tile_pack(const char* wl, int lane, WaveAcc& a) {
    constexpr int S = 16 / P2;                       // slabs per instruction
    constexpr int TR = 64 * Tile<T>::RPL;
    constexpr int steps = TR / (4 * S);
    const int f = lane & 15, q = lane >> 4;
    const int roff = q + 4 * (f / P2);               // this lane's row inside a 4 S-row step
    const T* xcol = reinterpret_cast<const T*>(wl + (f % P2) * kColStride) + roff;
    const T* ycol = reinterpret_cast<const T*>(wl + kSlotY * kColStride) + roff;
    const T* wcol = reinterpret_cast<const T*>(wl + kSlotW * kColStride) + roff;
    using Acc = typename Tile<T>::acc;
    Acc acc;
    if constexpr (sizeof(T) == 8) acc = Acc{a.d[0], a.d[1], a.d[2], a.d[3]};
    else acc = Acc{0, 0, 0, 0};
    T xy = 0, cs = 0, yy = 0, ys = 0, sw = 0;
    if constexpr (sizeof(T) == 8) {
        xy = a.xy; cs = a.cs; yy = a.yy; ys = a.ys; sw = a.sw;
    }
#pragma unroll
    for (int s = 0; s < steps; ++s) {
        const T x = xcol[4 * S * s], yv = ycol[4 * S * s];
        T xa = x;
        if constexpr (WEIGHTED) {
            const T wv = wcol[4 * S * s];
            xa = x * wv;
            yy = fma(wv * yv, yv, yy);
            ys = fma(wv, yv, ys);
            sw += wv;
        } else {
            yy = fma(yv, yv, yy);
            ys += yv;
        }
        acc = Tile<T>::mfma(xa, x, acc);
        xy = fma(xa, yv, xy);
        cs += xa;
    }
    if constexpr (sizeof(T) == 8) {
        a.d[0] = acc[0]; a.d[1] = acc[1]; a.d[2] = acc[2]; a.d[3] = acc[3];
        a.xy = xy; a.cs = cs; a.yy = yy; a.ys = ys; a.sw = sw;
    } else {
        a.d[0] += (double)acc[0]; a.d[1] += (double)acc[1]; a.d[2] += (double)acc[2]; a.d[3] += (double)acc[3];
        a.xy += (double)xy; a.cs += (double)cs; a.yy += (double)yy; a.ys += (double)ys; a.sw += (double)sw;
    }
}

// sum over the four row slots (lanes l, l^16, l^32, l^48).  gfx950's v_permlane16_swap / v_permlane32_swap exchange
// the odd rows (upper half) of one operand with the even rows (lower half) of the other: fed the same value twice they
// leave {own-or-partner, partner-or-own}, whose sum is v + xor-partner(v) on every lane -- no LDS round trip
// (__shfl_xor is ds_bpermute) and the same additions in the same order.
__device__ __forceinline__ double xor_sum_q(double v) {
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
    return v;
}

// write this wave's record (kPartStride doubles) : D tile, xy, cs, yy, ys, sw
template <typename T>
__device__ __forceinline__ void wave_record(const WaveAcc& a, int lane, double* rec) {
    const int j = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) rec[kPartD + Tile<T>::drow(lane, r) + 16 * j] = a.d[r];
    double xy = xor_sum_q(a.xy), cs = xor_sum_q(a.cs), yy = xor_sum_q(a.yy), ys = xor_sum_q(a.ys),
           sw = xor_sum_q(a.sw);
    if (lane < 16) {
        rec[kPartXY + lane] = xy;
        rec[kPartCS + lane] = cs;
    }
    if (lane == 0) {
        rec[kPartYY] = yy;
        rec[kPartYS] = ys;
        rec[kPartSW] = sw;
    }
}

// packed variant: the record keeps the full 16 x 16 tile and the 16 per-lane xy / cs sums (the finalize kernel folds the
// slabs); y'y, sum y and sum w are per-ROW sums, every slab saw different rows: one representative lane per slab is summed.
template <typename T, int P2>
__device__ __forceinline__ void wave_record_pack(const WaveAcc& a, int lane, double* rec) {
    const int j = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) rec[kPartD + Tile<T>::drow(lane, r) + 16 * j] = a.d[r];
    const double xy = xor_sum_q(a.xy), cs = xor_sum_q(a.cs);
    double yy = xor_sum_q(a.yy), ys = xor_sum_q(a.ys), sw = xor_sum_q(a.sw);
    const bool rep = (j % P2) == 0;  // first feature lane of its slab
    yy = rep ? yy : 0.0;
    ys = rep ? ys : 0.0;
    sw = rep ? sw : 0.0;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {  // over the 16 feature lanes of a row
        yy += __shfl_xor(yy, o);
        ys += __shfl_xor(ys, o);
        sw += __shfl_xor(sw, o);
    }
    if (lane < 16) {
        rec[kPartXY + lane] = xy;
        rec[kPartCS + lane] = cs;
    }
    if (lane == 0) {
        rec[kPartYY] = yy;
        rec[kPartYS] = ys;
        rec[kPartSW] = sw;
    }
}

__device__ __forceinline__ void zero_acc(WaveAcc& a) {
    a.d[0] = a.d[1] = a.d[2] = a.d[3] = 0.0;
    a.xy = a.cs = a.yy = a.ys = a.sw = 0.0;
}

template <typename T>
struct GroupRegs {
    T x[16][2];
    T y[2];
};

template <typename T>
__device__ __forceinline__ void load_group_tile(const ColPtrs<T>& cp, int p, int64_t r0, int64_t rend, int lane,
                                                GroupRegs<T>& g) {
    const int64_t ra = r0 + lane, rb = r0 + 64 + lane;
    const bool va = ra < rend, vb = rb < rend;
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (c < p) {
            g.x[c][0] = va ? cp.x[c][ra] : T(0);
            g.x[c][1] = vb ? cp.x[c][rb] : T(0);
        }
    g.y[0] = va ? cp.y[ra] : T(0);
    g.y[1] = vb ? cp.y[rb] : T(0);
}

template <typename T>
__device__ __forceinline__ void store_group_lds(char* wl, int p, int lane, const GroupRegs<T>& g) {
    // every feature slot is (re)written: slots >= p get exact zeros (the wave's result record
    // aliases the head of the tile region between groups)
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        T* col = reinterpret_cast<T*>(wl + c * kColStride);
        col[lane] = (c < p) ? g.x[c][0] : T(0);
        col[64 + lane] = (c < p) ? g.x[c][1] : T(0);
    }
    T* ycol = reinterpret_cast<T*>(wl + kSlotY * kColStride);
    ycol[lane] = g.y[0];
    ycol[64 + lane] = g.y[1];
}


}  // namespace pds
