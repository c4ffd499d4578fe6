// grouped_pred.hip -- per-row pred / resid of GROUPED fits: the second pass of `pl_lr_pred` (linear_regression.rs:782-806)
// for every group of a frame at once -- what `df.group_by(key).agg(pds.lin_reg(..., return_pred=True))` and
// `pds.lin_reg(..., return_pred=True).over(key)` make the reference do one group at a time
// (tests/test_linear_exprs.py:435-474, examples/basics.ipynb cells 16 / 18).
//
//   pred_r = x_r . beta_g(r) (+ b0_g(r)),  resid_r = y_r - pred_r,  row_null_r = (group g(r) is null)
//
// The frame is in GROUP ORDER (group g = rows [off[g], off[g+1]) -- the frame itself when its keys are ordered, the gathered
// copy of keyed.hip otherwise); `perm` (sorted position -> frame row, nullable) sends the results back to the frame's own
// row order, so a shuffled frame gets its predictions where its rows are.
//
// Layout: like pass2.hip -- lane = RPL consecutive rows (16-byte loads, 1 KiB coalesced per instruction), a wave owns a
// contiguous range of 64 RPL-row chunks and walks it upwards, so the group of a lane's rows only ever moves forward: one
// binary search when the wave starts, then `while (off[g + 1] <= r) ++g`.  The coefficient rows of the (two or three) groups
// a chunk touches are staged once per chunk in wave-private LDS.  HBM bound: reads N (p + 1) elements and the n_groups x p'
// coefficient block, writes 2 N elements + N bytes -- 3.0 ms at 1e8 x 16 f64 = the device's read + write copy rate.
#include "common.hpp"

#include <type_traits>

namespace pds {

namespace {

template <typename T>
struct GP16;
template <>
struct GP16<double> {
    typedef double type __attribute__((ext_vector_type(2), aligned(8)));
    static constexpr int RPL = 2;
};
template <>
struct GP16<float> {
    typedef float type __attribute__((ext_vector_type(4), aligned(4)));
    static constexpr int RPL = 4;
};

constexpr int kGpThreads = 256;
constexpr int kGpStage = 192;  // doubles of wave-private LDS for the coefficient rows of the groups a 64-lane chunk touches

// last group whose first row is <= r (empty groups are skipped over: off is non-decreasing)
__device__ __forceinline__ int64_t group_of_row(const int64_t* __restrict__ off, int64_t n_groups, int64_t r) {
    int64_t lo = 0, hi = n_groups;  // invariant: off[lo] <= r < off[hi]
    while (hi - lo > 1) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (off[mid] <= r) lo = mid;
        else hi = mid;
    }
    return lo;
}

// PC: feature count as a compile-time constant (1..16), 0 = run time (any width, column loop)
// MODE 0: rows in group order (offsets); 1: ... and results sent back through `perm`; 2: rows in ANY order, the group of a row looked
// up from its key (dense ids: group = rank[key - *kmin], keyed_partition.hip) -- the frame is read where it lies, nothing is
// permuted, and the n_groups x p' coefficient block is the only randomly read object (72 MB at 1e6 groups x 8 features: it lives
// in the L2s / the memory-side cache)
// MODE 3 (round 5): as MODE 2, but `coeffs` is an ID-indexed copy of the coefficient block (row id = key - *kmin; a null group's row
// starts with a NaN of payload 1; fill_coef_by_id_kernel below) -- the row's id IS its coefficient row, so rank[] and flags[] are not
// read: ONE randomly read object per row instead of three.
template <typename T> struct GpNullMark;
template <> struct GpNullMark<double> {
    static __device__ __forceinline__ double value() { return __builtin_bit_cast(double, 0x7ff8000000000001ull); }
    static __device__ __forceinline__ bool is(double v) { return __builtin_bit_cast(unsigned long long, v) == 0x7ff8000000000001ull; }
};
template <> struct GpNullMark<float> {
    static __device__ __forceinline__ float value() { return __builtin_bit_cast(float, 0x7fc00001u); }
    static __device__ __forceinline__ bool is(float v) { return __builtin_bit_cast(unsigned, v) == 0x7fc00001u; }
};
template <typename T>
__global__ __launch_bounds__(256) void fill_coef_by_id_kernel(const T* __restrict__ coeffs, const uint8_t* __restrict__ flags,
                                                              const unsigned* __restrict__ ids, int64_t n_groups, int pp, int stride,
                                                              T* __restrict__ table) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_groups * pp; i += (int64_t)gridDim.x * 256) {
        const int64_t g = i / pp;
        const int c = (int)(i - g * pp);
        T v = coeffs[i];
        if (c == 0 && flags && flags[g]) v = GpNullMark<T>::value();
        table[(size_t)ids[g] * stride + c] = v;
    }
}
template <typename T, int PC, int MODE>
__global__ __launch_bounds__(kGpThreads) void grouped_pred_kernel(const T* const* __restrict__ cols, int p_arg, int bias, int64_t n,
                                                                  const int64_t* __restrict__ off, int64_t n_groups,
                                                                  const T* __restrict__ coeffs, const uint8_t* __restrict__ flags,
                                                                  const uint32_t* __restrict__ perm /* MODE 2: rank */, T* __restrict__ pred,
                                                                  T* __restrict__ resid, uint8_t* __restrict__ row_null,
                                                                  const int64_t* __restrict__ keys, const int64_t* __restrict__ kmin) {
    constexpr bool PERM = MODE == 1;
    using V = typename GP16<T>::type;
    constexpr int RPL = GP16<T>::RPL;
    const int p = PC ? PC : p_arg;
    const int pp = p + bias;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (kGpThreads / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (kGpThreads / 64);
    const int64_t nchunk = (n + 64 * RPL - 1) / (64 * RPL);
    const int64_t c0 = nchunk * wave / nwaves, c1 = nchunk * (wave + 1) / nwaves;
    if (c0 >= c1) return;
    const T nanv = (T)__builtin_nan("");
    // coefficient rows of a chunk's groups, staged once per chunk (rows in group order only): per column step a lane then reads its
    // coefficient from LDS -- without the stage every column cost RPL more vector-memory instructions (51 instead of 17 per chunk
    // at 16 features; the kernel ran at 0.60 of the HBM peak against 0.82 for the single-regression residual pass)
    __shared__ T coef_stage[kGpThreads / 64][kGpStage];
    T* stage = coef_stage[threadIdx.x >> 6];
    const bool rn_vec = (reinterpret_cast<uintptr_t>(row_null) & (RPL - 1)) == 0;
    int64_t g[RPL];
    uint64_t kbase = 0;
    if constexpr (MODE >= 2) {
        kbase = (uint64_t)*kmin;
    } else {
        const int64_t r = c0 * 64 * RPL + (int64_t)lane * RPL;
        const int64_t g0 = group_of_row(off, n_groups, r < n ? r : n - 1);
#pragma unroll
        for (int e = 0; e < RPL; ++e) g[e] = g0;
    }
    const gptr<T> cy = as_global(cols[p]);
    for (int64_t ch = c0; ch < c1; ++ch) {
        const int64_t r0 = ch * 64 * RPL + (int64_t)lane * RPL;
        const bool full = r0 + RPL <= n;  // (lanes behind the last row stay in the loop -- clamped rows, masked stores: the chunk's
                                          // cross-lane steps below need the whole wave)
        if constexpr (MODE == 2) {
            // ---- the groups of the lane's rows: key -> dense id -> rank
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
                const int64_t r = (r0 + e < n) ? r0 + e : n - 1;
                g[e] = (int64_t)perm[(uint64_t)__builtin_nontemporal_load(keys + r) - kbase];
            }
        } else if constexpr (MODE == 3) {
            // ---- the lane's rows' ids: the coefficient table is indexed by them
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
                const int64_t r = (r0 + e < n) ? r0 + e : n - 1;
                g[e] = (int64_t)((uint64_t)__builtin_nontemporal_load(keys + r) - kbase);
            }
        } else {
            // ---- the groups of the lane's rows (monotone in r: forward steps only)
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
                const int64_t r = (r0 + e < n) ? r0 + e : n - 1;
                int64_t ge = e ? (g[e - 1] > g[e] ? g[e - 1] : g[e]) : g[0];
                while (ge + 1 < n_groups && off[ge + 1] <= r) ++ge;
                g[e] = ge;
            }
        }
        double acc[RPL];
        const T* brow[RPL];
        bool staged = false;
        if constexpr (MODE < 2) {
            const int64_t g_lo = __shfl(g[0], 0), g_hi = __shfl(g[RPL - 1], 63);
            const int64_t cnt = (g_hi - g_lo + 1) * pp;
            staged = cnt <= kGpStage;  // (wave-uniform)
            if (staged) {
                PDS_WAVE_LDS_SYNC();  // (the previous chunk's reads are done)
                const T* src = coeffs + g_lo * pp;
                for (int i = lane; i < (int)cnt; i += 64) stage[i] = src[i];
                PDS_WAVE_LDS_SYNC();
#pragma unroll
                for (int e = 0; e < RPL; ++e) brow[e] = stage + (g[e] - g_lo) * pp;
            }
        }
        if (!staged) {
            // (MODE 3: rows of the id-indexed table are padded to whole 64 / 128-byte lines -- one line per row instead of a 72-byte row
            //  straddling two; the stride travels in the n_groups argument, which this mode has no other use for)
            const int64_t cstride = (MODE == 3) ? n_groups : (int64_t)pp;
#pragma unroll
            for (int e = 0; e < RPL; ++e) brow[e] = coeffs + g[e] * cstride;
        }
#pragma unroll
        for (int e = 0; e < RPL; ++e) acc[e] = bias ? (double)brow[e][p] : 0.0;
        bool nl3[RPL];
        if constexpr (MODE == 3) {
#pragma unroll
            for (int e = 0; e < RPL; ++e) nl3[e] = GpNullMark<T>::is(brow[e][0]);
        }
        auto col_step = [&](int c) __attribute__((always_inline)) {
            const gptr<T> col = as_global(cols[c]);
            V v;
            if (full) {
                v = __builtin_nontemporal_load(reinterpret_cast<gptr<V>>(col + r0));
            } else {
#pragma unroll
                for (int e = 0; e < RPL; ++e) v[e] = (r0 + e < n) ? col[r0 + e] : T(0);
            }
#pragma unroll
            for (int e = 0; e < RPL; ++e) acc[e] = fma((double)v[e], (double)brow[e][c], acc[e]);
        };
        if constexpr (PC != 0) {
#pragma unroll
            for (int c = 0; c < PC; ++c) col_step(c);
        } else {
#pragma unroll 4
            for (int c = 0; c < p; ++c) col_step(c);
        }
        V yv;
        if (full) {
            yv = __builtin_nontemporal_load(reinterpret_cast<gptr<V>>(cy + r0));
        } else {
#pragma unroll
            for (int e = 0; e < RPL; ++e) yv[e] = (r0 + e < n) ? cy[r0 + e] : T(0);
        }
        V pv, rv;
        uint8_t nl[RPL];
#pragma unroll
        for (int e = 0; e < RPL; ++e) {
            if constexpr (MODE == 3) nl[e] = nl3[e] ? (uint8_t)1 : (uint8_t)0;
            else nl[e] = flags ? flags[g[e]] : (uint8_t)0;
            const T pr = nl[e] ? nanv : (T)acc[e];
            pv[e] = pr;
            rv[e] = nl[e] ? nanv : (T)((double)yv[e] - (double)pr);
        }
        if constexpr (PERM) {
#pragma unroll
            for (int e = 0; e < RPL; ++e)
                if (r0 + e < n) {
                    const int64_t dst = (int64_t)perm[r0 + e];
                    if (pred) pred[dst] = pv[e];
                    if (resid) resid[dst] = rv[e];
                    if (row_null) row_null[dst] = nl[e];
                }
        } else if (full) {
            if (pred) *reinterpret_cast<V*>(pred + r0) = pv;
            if (resid) *reinterpret_cast<V*>(resid + r0) = rv;
            if (row_null) {
                if (rn_vec) {  // one RPL-byte store per lane instead of RPL byte stores
                    if constexpr (RPL == 2) *reinterpret_cast<uint16_t*>(row_null + r0) = (uint16_t)(nl[0] | (nl[1] << 8));
                    else *reinterpret_cast<uint32_t*>(row_null + r0) = (uint32_t)nl[0] | ((uint32_t)nl[1] << 8) | ((uint32_t)nl[2] << 16) | ((uint32_t)nl[3] << 24);
                } else {
#pragma unroll
                    for (int e = 0; e < RPL; ++e) row_null[r0 + e] = nl[e];
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < RPL; ++e)
                if (r0 + e < n) {
                    if (pred) pred[r0 + e] = pv[e];
                    if (resid) resid[r0 + e] = rv[e];
                    if (row_null) row_null[r0 + e] = nl[e];
                }
        }
    }
}

template <typename T, int MODE>
void launch_pc(int p, dim3 g, hipStream_t st, const T* const* cols, int bias, int64_t n, const int64_t* off, int64_t ng,
               const T* co, const uint8_t* fl, const uint32_t* perm, T* pred, T* resid, uint8_t* rn, const int64_t* keys = nullptr,
               const int64_t* kmin = nullptr) {
#define PDS_GP_CASE(PCV)                                                                                                         \
    case PCV:                                                                                                                    \
        hipLaunchKernelGGL((grouped_pred_kernel<T, PCV, MODE>), g, dim3(kGpThreads), 0, st, cols, p, bias, n, off, ng, co, fl, perm, \
                           pred, resid, rn, keys, kmin);                                                                         \
        break;
    switch (p) {
        PDS_GP_CASE(1)
        PDS_GP_CASE(2)
        PDS_GP_CASE(3)
        PDS_GP_CASE(4)
        PDS_GP_CASE(5)
        PDS_GP_CASE(6)
        PDS_GP_CASE(7)
        PDS_GP_CASE(8)
        PDS_GP_CASE(9)
        PDS_GP_CASE(10)
        PDS_GP_CASE(11)
        PDS_GP_CASE(12)
        PDS_GP_CASE(13)
        PDS_GP_CASE(14)
        PDS_GP_CASE(15)
        PDS_GP_CASE(16)
        default:
            hipLaunchKernelGGL((grouped_pred_kernel<T, 0, MODE>), g, dim3(kGpThreads), 0, st, cols, p, bias, n, off, ng, co, fl, perm,
                               pred, resid, rn, keys, kmin);
    }
#undef PDS_GP_CASE
}

}  // namespace

// d_cols: device table in kernel order (x_0 .. x_{p-1}, y); d_off: n_groups + 1 device offsets; d_coeffs n_groups x p' row-major;
// d_flags (nullable) one byte per group; d_perm (nullable) sorted position -> frame row; outputs (each nullable) in frame order
template <typename T>
int launch_grouped_pred(pds_ctx* ctx, const T* const* d_cols, int n_feat, int bias, int64_t n_rows, const int64_t* d_off,
                        int64_t n_groups, const T* d_coeffs, const uint8_t* d_flags, const uint32_t* d_perm, T* d_pred, T* d_resid,
                        uint8_t* d_row_null) {
    if (n_rows <= 0 || n_groups <= 0) return PDS_OK;
    KernelTimer timer(ctx, kKindPass2);
    constexpr int RPL = GP16<T>::RPL;
    const int64_t nchunk = (n_rows + 64 * RPL - 1) / (64 * RPL);
    const int nb = (int)std::min<int64_t>(std::max<int64_t>((nchunk + 3) / 4, 1), (int64_t)ctx->num_cus * 8);
    if (d_perm)
        launch_pc<T, 1>(n_feat, dim3(nb), ctx->stream, d_cols, bias, n_rows, d_off, n_groups, d_coeffs, d_flags, d_perm, d_pred,
                           d_resid, d_row_null);
    else
        launch_pc<T, 0>(n_feat, dim3(nb), ctx->stream, d_cols, bias, n_rows, d_off, n_groups, d_coeffs, d_flags, d_perm, d_pred,
                            d_resid, d_row_null);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
// rows in any order, dense integer keys: group of row r = d_rank[keys[r] - *d_kmin] (the partition route's id -> group table)
template <typename T>
int launch_grouped_pred_by_id(pds_ctx* ctx, const T* const* d_cols, int n_feat, int bias, int64_t n_rows, const int64_t* d_keys,
                              const int64_t* d_kmin, const uint32_t* d_rank, int64_t n_groups, const T* d_coeffs, const uint8_t* d_flags,
                              T* d_pred, T* d_resid, uint8_t* d_row_null) {
    if (n_rows <= 0 || n_groups <= 0) return PDS_OK;
    KernelTimer timer(ctx, kKindPass2);
    constexpr int RPL = GP16<T>::RPL;
    const int64_t nchunk = (n_rows + 64 * RPL - 1) / (64 * RPL);
    const int nb = (int)std::min<int64_t>(std::max<int64_t>((nchunk + 3) / 4, 1), (int64_t)ctx->num_cus * 8);
    launch_pc<T, 2>(n_feat, dim3(nb), ctx->stream, d_cols, bias, n_rows, nullptr, n_groups, d_coeffs, d_flags, d_rank, d_pred, d_resid,
                    d_row_null, d_keys, d_kmin);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
// the same with an id-indexed coefficient table (d_table: n_ids x p' values of workspace, filled here from the compact block through
// d_ids = dense id of group g): one randomly read object per row
template <typename T>
int launch_grouped_pred_by_id_table(pds_ctx* ctx, const T* const* d_cols, int n_feat, int bias, int64_t n_rows, const int64_t* d_keys,
                                    const int64_t* d_kmin, const uint32_t* d_ids, int64_t n_groups, const T* d_coeffs, const uint8_t* d_flags,
                                    T* d_table, T* d_pred, T* d_resid, uint8_t* d_row_null) {
    if (n_rows <= 0 || n_groups <= 0) return PDS_OK;
    KernelTimer timer(ctx, kKindPass2);
    const int pp = n_feat + bias;
    const int fb = (int)std::min<int64_t>((n_groups * pp + 255) / 256, (int64_t)ctx->num_cus * 16);
    const int stride = grouped_pred_table_stride<T>(pp);
    hipLaunchKernelGGL((fill_coef_by_id_kernel<T>), dim3(fb), dim3(256), 0, ctx->stream, d_coeffs, d_flags, d_ids, n_groups, pp, stride, d_table);
    constexpr int RPL = GP16<T>::RPL;
    const int64_t nchunk = (n_rows + 64 * RPL - 1) / (64 * RPL);
    const int nb = (int)std::min<int64_t>(std::max<int64_t>((nchunk + 3) / 4, 1), (int64_t)ctx->num_cus * 8);
    launch_pc<T, 3>(n_feat, dim3(nb), ctx->stream, d_cols, bias, n_rows, nullptr, (int64_t)stride, (const T*)d_table, nullptr, nullptr, d_pred, d_resid,
                    d_row_null, d_keys, d_kmin);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
template int launch_grouped_pred_by_id_table<double>(pds_ctx*, const double* const*, int, int, int64_t, const int64_t*, const int64_t*,
                                                     const uint32_t*, int64_t, const double*, const uint8_t*, double*, double*, double*, uint8_t*);
template int launch_grouped_pred_by_id_table<float>(pds_ctx*, const float* const*, int, int, int64_t, const int64_t*, const int64_t*,
                                                    const uint32_t*, int64_t, const float*, const uint8_t*, float*, float*, float*, uint8_t*);
template int launch_grouped_pred_by_id<double>(pds_ctx*, const double* const*, int, int, int64_t, const int64_t*, const int64_t*,
                                               const uint32_t*, int64_t, const double*, const uint8_t*, double*, double*, uint8_t*);
template int launch_grouped_pred_by_id<float>(pds_ctx*, const float* const*, int, int, int64_t, const int64_t*, const int64_t*,
                                              const uint32_t*, int64_t, const float*, const uint8_t*, float*, float*, uint8_t*);
template int launch_grouped_pred<double>(pds_ctx*, const double* const*, int, int, int64_t, const int64_t*, int64_t, const double*,
                                         const uint8_t*, const uint32_t*, double*, double*, uint8_t*);
template int launch_grouped_pred<float>(pds_ctx*, const float* const*, int, int, int64_t, const int64_t*, int64_t, const float*,
                                        const uint8_t*, const uint32_t*, float*, float*, uint8_t*);

}  // namespace pds
