// layout.hip -- row-major matrix -> the contiguous column buffers every kernel of this library streams.
// The reference's pyclass route (PyLR / PyElasticNet / PyOnlineLR, src/pymodels/py_lr.rs) reads a row-major NumPy matrix
// through a strided faer MatRef (src/pymodels/numpy_faer.rs:10-66).  Here the matrix is transposed ONCE on the device:
// 64-row x 32-column tiles through LDS (reads: 256 contiguous bytes per row piece, writes: 512 contiguous bytes per column
// piece, LDS stride 33: conflict free both ways).  A host matrix crosses PCIe as contiguous row chunks (one copy per chunk
// instead of one strided host gather per column) into a chunk-sized staging buffer.
#include "common.hpp"

namespace pds {

constexpr int kTrRows = 64, kTrCols = 32;

template <typename T>
__global__ __launch_bounds__(256) void rows_to_cols_kernel(const T* __restrict__ X, int64_t ld, int64_t n_rows, int n_cols,
                                                           T* __restrict__ out, int64_t col_stride, int64_t out_row0) {
    __shared__ T tile[kTrRows][kTrCols + 1];
    const int64_t r0 = (int64_t)blockIdx.x * kTrRows;
    const int c0 = blockIdx.y * kTrCols;
    const int rows = (int)((n_rows - r0 < kTrRows) ? n_rows - r0 : kTrRows);
    const int cols = (n_cols - c0 < kTrCols) ? n_cols - c0 : kTrCols;
    for (int e = threadIdx.x; e < kTrRows * kTrCols; e += 256) {
        const int r = e / kTrCols, c = e % kTrCols;
        if (r < rows && c < cols) tile[r][c] = X[(r0 + r) * ld + c0 + c];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kTrRows * kTrCols; e += 256) {
        const int c = e / kTrRows, r = e % kTrRows;
        if (r < rows && c < cols) out[(int64_t)(c0 + c) * col_stride + out_row0 + r0 + r] = tile[r][c];
    }
}

template <typename T>
int launch_rows_to_cols(pds_ctx* ctx, const T* d_X, int64_t ld, int64_t n_rows, int n_cols, T* d_out, int64_t col_stride,
                        int64_t out_row0) {
    if (n_rows <= 0 || n_cols <= 0) return PDS_OK;
    const dim3 grid((unsigned)((n_rows + kTrRows - 1) / kTrRows), (unsigned)((n_cols + kTrCols - 1) / kTrCols));
    hipLaunchKernelGGL((rows_to_cols_kernel<T>), grid, dim3(256), 0, ctx->stream, d_X, ld, n_rows, n_cols, d_out, col_stride, out_row0);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
template int launch_rows_to_cols<double>(pds_ctx*, const double*, int64_t, int64_t, int, double*, int64_t, int64_t);
template int launch_rows_to_cols<float>(pds_ctx*, const float*, int64_t, int64_t, int, float*, int64_t, int64_t);

}  // namespace pds
