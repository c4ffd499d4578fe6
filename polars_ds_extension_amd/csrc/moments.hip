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

// ---- full-tile load: lane reads RPL consecutive rows of every column (16 B, coalesced 1 KiB/instr)
template <typename T, bool WEIGHTED>
__device__ __forceinline__ void load_full_tile(const T* const* __restrict__ cols, int p, int64_t row,
                                               TileRegs<T>& r) {
    using V = typename Tile<T>::vec;
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (c < p) r.x[c] = *reinterpret_cast<const V*>(cols[c] + row);
    r.y = *reinterpret_cast<const V*>(cols[p] + row);
    if (WEIGHTED) r.w = *reinterpret_cast<const V*>(cols[p + 1] + row);
}

// ---- guarded load for the ragged last tile (rows >= n contribute exact zeros)
template <typename T, bool WEIGHTED>
__device__ __forceinline__ void load_tail_tile(const T* const* __restrict__ cols, int p, int64_t row,
                                               int64_t n, TileRegs<T>& r) {
    constexpr int RPL = Tile<T>::RPL;
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (c < p) {
#pragma unroll
            for (int e = 0; e < RPL; ++e) r.x[c][e] = (row + e < n) ? cols[c][row + e] : T(0);
        }
#pragma unroll
    for (int e = 0; e < RPL; ++e) r.y[e] = (row + e < n) ? cols[p][row + e] : T(0);
    if (WEIGHTED) {
#pragma unroll
        for (int e = 0; e < RPL; ++e) r.w[e] = (row + e < n) ? cols[p + 1][row + e] : T(0);
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
template <typename T, bool WEIGHTED>
__device__ __forceinline__ void consume_tile(const char* wl, int lane, int steps, WaveAcc& a) {
    const int f = lane & 15, q = lane >> 4;
    const T* xcol = reinterpret_cast<const T*>(wl + f * kColStride) + q;
    const T* ycol = reinterpret_cast<const T*>(wl + kSlotY * kColStride) + q;
    const T* wcol = reinterpret_cast<const T*>(wl + kSlotW * kColStride) + q;
    if constexpr (sizeof(T) == 8) {
        d4 acc = {a.d[0], a.d[1], a.d[2], a.d[3]};
        double xy = a.xy, cs = a.cs, yy = a.yy, ys = a.ys, sw = a.sw;
#pragma unroll 8
        for (int s = 0; s < steps; ++s) {
            double x = xcol[4 * s];
            double yv = ycol[4 * s];
            double xa = x;
            if (WEIGHTED) {
                double wv = wcol[4 * s];
                xa = x * wv;
                yy = fma(wv * yv, yv, yy);
                ys = fma(wv, yv, ys);
                sw += wv;
            } else {
                yy = fma(yv, yv, yy);
                ys += yv;
            }
            acc = Tile<double>::mfma(xa, x, acc);
            xy = fma(xa, yv, xy);
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
                yy = fmaf(wv * yv, yv, yy);
                ys = fmaf(wv, yv, ys);
                sw += wv;
            } else {
                yy = fmaf(yv, yv, yy);
                ys += yv;
            }
            acc = Tile<float>::mfma(xa, x, acc);
            xy = fmaf(xa, yv, xy);
            cs += xa;
        }
        a.d[0] += (double)acc[0]; a.d[1] += (double)acc[1]; a.d[2] += (double)acc[2]; a.d[3] += (double)acc[3];
        a.xy += (double)xy; a.cs += (double)cs; a.yy += (double)yy; a.ys += (double)ys; a.sw += (double)sw;
    }
}

__device__ __forceinline__ double xor_sum_q(double v) {  // sum over the four row slots (lanes l, l^16, l^32, l^48)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
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

__device__ __forceinline__ void zero_acc(WaveAcc& a) {
    a.d[0] = a.d[1] = a.d[2] = a.d[3] = 0.0;
    a.xy = a.cs = a.yy = a.ys = a.sw = 0.0;
}

// =============================================================================================
// single big system: grid-stride over 128-row (f64) / 256-row (f32) tiles, register prefetch of the
// next tile while the matrix core chews the current one.
// =============================================================================================
template <typename T, bool WEIGHTED>
__global__ __launch_bounds__(256, 2) void moments_small_kernel(const T* const* __restrict__ cols, int p,
                                                               int64_t n, double* __restrict__ partials) {
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

    WaveAcc acc;
    zero_acc(acc);
    const int64_t nfull = n / TR;
    const int64_t wid = (int64_t)blockIdx.x * kWaves + wave, nw = (int64_t)gridDim.x * kWaves;
    TileRegs<T> regs;
    int64_t t = wid;
    if (t < nfull) load_full_tile<T, WEIGHTED>(cols, p, t * TR + lane * RPL, regs);
    for (; t < nfull; t += nw) {
        store_tile_lds<T, WEIGHTED>(wl, p, lane, regs);
        const int64_t tn = t + nw;
        if (tn < nfull) load_full_tile<T, WEIGHTED>(cols, p, tn * TR + lane * RPL, regs);
        consume_tile<T, WEIGHTED>(wl, lane, TR / 4, acc);
    }
    if (nfull * TR < n && (nfull % nw) == wid) {  // ragged tail: exactly one wave
        load_tail_tile<T, WEIGHTED>(cols, p, nfull * TR + lane * RPL, n, regs);
        store_tile_lds<T, WEIGHTED>(wl, p, lane, regs);
        consume_tile<T, WEIGHTED>(wl, lane, TR / 4, acc);
    }

    // block reduction through LDS (tile storage is dead now)
    __syncthreads();
    double* recs = reinterpret_cast<double*>(smem);
    wave_record<T>(acc, lane, recs + wave * kPartStride);
    __syncthreads();
    for (int e = threadIdx.x; e < kPartStride; e += blockDim.x) {
        double s = 0.0;
        if (e <= kPartSW) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s += recs[w * kPartStride + e];
        }
        partials[(int64_t)blockIdx.x * kPartStride + e] = s;
    }
}

// fixed-order reduction of the per-block partials and assembly of the (p+2)x(p+2) moment matrix.
// One 64-lane block per record element: lane l sums blocks l, l+64, ... in order, then a fixed
// butterfly combines the lanes -- deterministic, and ~2 us instead of a 512-long dependent load chain.
template <typename T>
__global__ __launch_bounds__(64) void moments_finalize_kernel(const double* __restrict__ partials, int nblocks,
                                                              int p, double n_rows, int weighted,
                                                              T* __restrict__ out) {
    const int e = blockIdx.x;  // element of the partial record, 0 .. kPartSW
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partials[(int64_t)b * kPartStride + e];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
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
    } else if (e == kPartSW) {
        out[p + p * q] = (T)(weighted ? s : n_rows);
    }
}

// =============================================================================================
// grouped: one wave per group at a time (rows of a group are contiguous, group starts are only
// element-aligned -> 8/4-byte lane loads, still 512/256 B coalesced per instruction)
// =============================================================================================
template <typename T>
struct GroupRegs {
    T x[16][2];
    T y[2];
};

template <typename T>
__device__ __forceinline__ void load_group_tile(const T* const* __restrict__ cols, int p, int64_t r0,
                                                int64_t rend, int lane, GroupRegs<T>& g) {
    const int64_t ra = r0 + lane, rb = r0 + 64 + lane;
    const bool va = ra < rend, vb = rb < rend;
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (c < p) {
            g.x[c][0] = va ? cols[c][ra] : T(0);
            g.x[c][1] = vb ? cols[c][rb] : T(0);
        }
    g.y[0] = va ? cols[p][ra] : T(0);
    g.y[1] = vb ? cols[p][rb] : T(0);
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

template <typename T>
__global__ __launch_bounds__(256, 2) void grouped_moments_kernel(const T* const* __restrict__ cols, int p,
                                                                 const int64_t* __restrict__ offsets,
                                                                 int64_t n_groups, T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* wl = smem + wave * kWaveLds;
    double* rec = reinterpret_cast<double*>(wl);  // aliases the (dead) tile between two groups
    const int q = p + 2;
    const int64_t wid = (int64_t)blockIdx.x * kWaves + wave, nw = (int64_t)gridDim.x * kWaves;
    GroupRegs<T> regs;
    int64_t g = wid;
    int64_t r0 = 0, rend = 0;
    if (g < n_groups) {
        r0 = offsets[g];
        rend = offsets[g + 1];
        load_group_tile<T>(cols, p, r0, rend, lane, regs);
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
            r0 = offsets[gn];
            rend = offsets[gn + 1];
            load_group_tile<T>(cols, p, r0, rend, lane, regs);
        }
        {
            const int rows = (int)(ng < 128 ? ng : 128);
            consume_tile<T, false>(wl, lane, (rows + 3) >> 2, acc);
        }
        while (done < ng) {  // long groups: stream the rest 128 rows at a time
            load_group_tile<T>(cols, p, gr0 + done, grend, lane, regs);
            store_group_lds<T>(wl, p, lane, regs);
            const int64_t left = ng - done;
            const int rows = (int)(left < 128 ? left : 128);
            consume_tile<T, false>(wl, lane, (rows + 3) >> 2, acc);
            done += 128;
            if (done >= ng && gn < n_groups) {
                r0 = offsets[gn];
                rend = offsets[gn + 1];
                load_group_tile<T>(cols, p, r0, rend, lane, regs);
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
                   T* d_moments) {
    if (n_feat < 1 || n_feat > kMaxFeatSmall)
        return fail(PDS_ERR_UNSUPPORTED, "moments: this build handles 1..16 features in the MFMA tile kernel");
    constexpr int TR = 64 * Tile<T>::RPL;
    int64_t ntiles = (n_rows + TR - 1) / TR;
    int64_t want = (ntiles + kWaves - 1) / kWaves;
    int nblocks = (int)std::min<int64_t>(std::max<int64_t>(want, 1), (int64_t)ctx->num_cus * 2);
    double* partials = ctx->partials;  // sized for 8 blocks per CU at context creation
    KernelTimer timer(ctx, kKindMoments);
    size_t lds = (size_t)kWaves * kWaveLds;
    if (weighted)
        hipLaunchKernelGGL((moments_small_kernel<T, true>), dim3(nblocks), dim3(256), lds, ctx->stream, dc.d_ptrs,
                           n_feat, n_rows, partials);
    else
        hipLaunchKernelGGL((moments_small_kernel<T, false>), dim3(nblocks), dim3(256), lds, ctx->stream, dc.d_ptrs,
                           n_feat, n_rows, partials);
    hipLaunchKernelGGL((moments_finalize_kernel<T>), dim3(kPartSW + 1), dim3(64), 0, ctx->stream, partials, nblocks, n_feat,
                       (double)n_rows, weighted ? 1 : 0, d_moments);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

template <typename T>
int launch_grouped_moments(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, const int64_t* d_offsets,
                           int64_t n_groups, T* d_moments) {
    if (n_feat < 1 || n_feat > kMaxFeatSmall)
        return fail(PDS_ERR_UNSUPPORTED, "grouped moments: 1..16 features supported");
    if (n_groups <= 0) return PDS_OK;
    int64_t want = (n_groups + kWaves - 1) / kWaves;
    int nblocks = (int)std::min<int64_t>(want, (int64_t)ctx->num_cus * 2);
    size_t lds = (size_t)kWaves * kWaveLds;
    KernelTimer timer(ctx, kKindGroupedMoments);
    hipLaunchKernelGGL((grouped_moments_kernel<T>), dim3(nblocks), dim3(256), lds, ctx->stream, dc.d_ptrs, n_feat,
                       d_offsets, n_groups, d_moments);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

template int launch_moments<double>(pds_ctx*, const DeviceCols<double>&, int, int64_t, bool, double*);
template int launch_moments<float>(pds_ctx*, const DeviceCols<float>&, int, int64_t, bool, float*);
template int launch_grouped_moments<double>(pds_ctx*, const DeviceCols<double>&, int, const int64_t*, int64_t,
                                            double*);
template int launch_grouped_moments<float>(pds_ctx*, const DeviceCols<float>&, int, const int64_t*, int64_t,
                                           float*);

}  // namespace pds
