// nulls.hip -- the null handling of `series_to_mat_for_lr` (/root/reference/src/num_ext/linear_regression.rs
// :151-267) on the device, driven by the Arrow validity bitmaps of the input columns.
//
// The reference builds a Polars DataFrame, filters / fills it, and copies the survivors into one column-major
// Vec.  Here the same policies produce compacted (or filled) column buffers in HBM that every other kernel then
// reads unchanged:
//   RAISE   any null -> error "Nulls found in data"                                        (:198)
//   SKIP    keep rows where every input is valid (mask = AND of the bitmaps), drop the rest  (:199-206)
//   FILL v  features: null -> v; if the target has nulls those rows are dropped              (:207-227)
//   IGNORE  null -> NaN, nothing dropped (the NaNs then poison the fit exactly as in the reference, :194-197)
// Stream compaction is hipCUB's DeviceSelect::Flagged (one pass per column); the exclusive scan of the keep
// flags is kept for scattering per-row outputs (pred / resid) back to their original positions.
#include <hipcub/hipcub.hpp>

#include <limits>

#include "common.hpp"

namespace pds {

__device__ __forceinline__ bool bit_at(const uint8_t* __restrict__ bm, int64_t i) {
    return bm == nullptr || ((bm[i >> 3] >> (i & 7)) & 1);
}

// keep[r] = AND over the selected columns of their validity bit; also counts nulls (any column)
__global__ __launch_bounds__(256) void keep_mask_kernel(const uint8_t* const* __restrict__ bitmaps,
                                                        const int64_t* __restrict__ bit_off, int ncols, int and_all,
                                                        int64_t n, uint8_t* __restrict__ keep,
                                                        unsigned long long* __restrict__ null_count) {
    unsigned long long local = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        bool k = true, any_null = false;
        for (int c = 0; c < ncols; ++c) {
            const bool v = bit_at(bitmaps[c], r + bit_off[c]);
            any_null |= !v;
            if (and_all || c == 0) k = k && v;  // column 0 is the target
        }
        keep[r] = k ? 1 : 0;
        local += any_null ? 1 : 0;
    }
    for (int o = 32; o >= 1; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(null_count, local);
}

template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(const T* __restrict__ in, const uint8_t* __restrict__ bm,
                                                   int64_t bit_off, int64_t n, T fill, T* __restrict__ out) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
        out[r] = bit_at(bm, r + bit_off) ? in[r] : fill;
}

template <typename T>
__global__ __launch_bounds__(256) void expand_kernel(const T* __restrict__ compact, const uint8_t* __restrict__ keep,
                                                     const int64_t* __restrict__ rank, int64_t n, T* __restrict__ out,
                                                     uint8_t* __restrict__ valid) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const bool k = keep[r];
        out[r] = k ? compact[rank[r]] : (T)__builtin_nan("");
        if (valid) valid[r] = k ? 1 : 0;
    }
}

struct KeepToI64 {
    __host__ __device__ int64_t operator()(uint8_t v) const { return (int64_t)v; }
};

// cols_dev: device-resident columns [y, x1..xp]; bitmaps (host array of DEVICE pointers or nullptr), bit offsets.
template <typename T>
int apply_null_policy(pds_ctx* ctx, const std::vector<const T*>& cols_dev, const std::vector<const uint8_t*>& bm_dev,
                      const std::vector<int64_t>& bit_off, int64_t n_rows, int policy, T fill_value,
                      NullPrepared<T>& out) {
    const int nc = (int)cols_dev.size();
    out.cols = cols_dev;
    out.n_kept = n_rows;
    out.dropped = false;
    bool any_bitmap = false;
    for (auto b : bm_dev) any_bitmap |= b != nullptr;
    if (!any_bitmap) return PDS_OK;
    const int nblocks = (int)std::min<int64_t>((n_rows + 255) / 256, (int64_t)ctx->num_cus * 8);
    // device tables
    const uint8_t** d_bms = reinterpret_cast<const uint8_t**>(ws_take(ctx, sizeof(void*) * nc));
    int64_t* d_off = reinterpret_cast<int64_t*>(ws_take(ctx, sizeof(int64_t) * nc));
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(ws_take(ctx, 16));
    uint8_t* d_keep = reinterpret_cast<uint8_t*>(ws_take(ctx, (size_t)n_rows));
    PDS_HIP_CHECK(hipMemcpyAsync(d_bms, bm_dev.data(), sizeof(void*) * nc, hipMemcpyHostToDevice, ctx->stream));
    PDS_HIP_CHECK(hipMemcpyAsync(d_off, bit_off.data(), sizeof(int64_t) * nc, hipMemcpyHostToDevice, ctx->stream));
    PDS_HIP_CHECK(hipMemsetAsync(d_cnt, 0, 16, ctx->stream));
    const int and_all = policy == PDS_NULL_SKIP ? 1 : 0;
    hipLaunchKernelGGL(keep_mask_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, d_bms, d_off, nc, and_all, n_rows, d_keep,
                       d_cnt);
    unsigned long long h_cnt = 0;
    PDS_HIP_CHECK(hipMemcpyAsync(&h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (h_cnt == 0) return PDS_OK;  // bitmaps present but no null inside the slice: the fast path
    if (policy == PDS_NULL_RAISE) return fail(PDS_ERR_NULLS, "Nulls found in data");
    // ---- fill (FILL: features only; IGNORE: every column gets NaN)
    std::vector<const T*> work = cols_dev;
    if (policy == PDS_NULL_FILL || policy == PDS_NULL_IGNORE) {
        const T nanv = std::numeric_limits<T>::quiet_NaN();
        for (int c = 0; c < nc; ++c) {
            if (!bm_dev[c]) continue;
            if (policy == PDS_NULL_FILL && c == 0) continue;  // target nulls are dropped, not filled
            T* dst = reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T)));
            hipLaunchKernelGGL((fill_kernel<T>), dim3(nblocks), dim3(256), 0, ctx->stream, cols_dev[c], bm_dev[c], bit_off[c],
                               n_rows, policy == PDS_NULL_FILL ? fill_value : nanv, dst);
            work[c] = dst;
        }
    }
    const bool drop = policy == PDS_NULL_SKIP || (policy == PDS_NULL_FILL && bm_dev[0] != nullptr);
    if (!drop) {
        out.cols = work;
        return PDS_OK;
    }
    // ---- compaction: rank = exclusive scan of keep, then DeviceSelect::Flagged per column
    int64_t* d_rank = reinterpret_cast<int64_t*>(ws_take(ctx, (size_t)n_rows * sizeof(int64_t)));
    int64_t* d_nsel = reinterpret_cast<int64_t*>(ws_take(ctx, 16));
    size_t tmp_scan = 0, tmp_sel = 0;
    hipcub::TransformInputIterator<int64_t, KeepToI64, const uint8_t*> keep_it(d_keep, KeepToI64());
    PDS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scan, keep_it, d_rank, n_rows, ctx->stream));
    PDS_HIP_CHECK(hipcub::DeviceSelect::Flagged(nullptr, tmp_sel, (const T*)nullptr, (const uint8_t*)nullptr, (T*)nullptr,
                                                d_nsel, n_rows, ctx->stream));
    void* d_tmp = ws_take(ctx, std::max(tmp_scan, tmp_sel) + 256);
    size_t tb = tmp_scan;
    PDS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tb, keep_it, d_rank, n_rows, ctx->stream));
    for (int c = 0; c < nc; ++c) {
        T* dst = reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T)));
        tb = tmp_sel;
        PDS_HIP_CHECK(hipcub::DeviceSelect::Flagged(d_tmp, tb, work[c], d_keep, dst, d_nsel, n_rows, ctx->stream));
        work[c] = dst;
    }
    int64_t h_nsel = 0;
    PDS_HIP_CHECK(hipMemcpyAsync(&h_nsel, d_nsel, sizeof(h_nsel), hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    out.cols = work;
    out.n_kept = h_nsel;
    out.dropped = true;
    out.d_keep = d_keep;
    out.d_rank = d_rank;
    return PDS_OK;
}

template <typename T>
int expand_rows(pds_ctx* ctx, const T* d_compact, const uint8_t* d_keep, const int64_t* d_rank, int64_t n_rows, T* d_out,
                uint8_t* d_valid) {
    const int nblocks = (int)std::min<int64_t>((n_rows + 255) / 256, (int64_t)ctx->num_cus * 8);
    hipLaunchKernelGGL((expand_kernel<T>), dim3(nblocks), dim3(256), 0, ctx->stream, d_compact, d_keep, d_rank, n_rows, d_out,
                       d_valid);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

// group g = rows [off[g], off[g+1]) of the original frame -> the same group in the compacted frame
__global__ __launch_bounds__(256) void remap_offsets_kernel(const int64_t* __restrict__ off, int64_t n_groups,
                                                            const int64_t* __restrict__ rank, int64_t n_rows, int64_t n_kept,
                                                            int64_t* __restrict__ out) {
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g <= n_groups; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = off[g];
        out[g] = r < n_rows ? rank[r] : n_kept;  // rank = number of kept rows in front of row r
    }
}

int remap_group_offsets(pds_ctx* ctx, const int64_t* d_off, int64_t n_groups, const int64_t* d_rank, int64_t n_rows,
                        int64_t n_kept, int64_t* d_out) {
    const int nblocks = (int)std::min<int64_t>((n_groups + 256) / 256, (int64_t)ctx->num_cus * 8);
    hipLaunchKernelGGL(remap_offsets_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, d_off, n_groups, d_rank, n_rows, n_kept,
                       d_out);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

size_t null_policy_workspace(int n_cols, int64_t n_rows, size_t elem) {
    // keep + rank + (fill + compact) copies of every column + cub temporaries
    return (size_t)n_rows * (1 + 8 + 2 * (size_t)n_cols * elem) + ((size_t)n_rows / 64 + 65536) * 8 + (1 << 20);
}

template int apply_null_policy<double>(pds_ctx*, const std::vector<const double*>&, const std::vector<const uint8_t*>&,
                                       const std::vector<int64_t>&, int64_t, int, double, NullPrepared<double>&);
template int apply_null_policy<float>(pds_ctx*, const std::vector<const float*>&, const std::vector<const uint8_t*>&,
                                      const std::vector<int64_t>&, int64_t, int, float, NullPrepared<float>&);
template int expand_rows<double>(pds_ctx*, const double*, const uint8_t*, const int64_t*, int64_t, double*, uint8_t*);
template int expand_rows<float>(pds_ctx*, const float*, const uint8_t*, const int64_t*, int64_t, float*, uint8_t*);

}  // namespace pds
