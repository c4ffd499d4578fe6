// capi_staging.hpp -- host frames in row chunks (O(chunk) HBM) and row-major -> column transposition
// Part of the one translation unit capi.hip (included there, inside namespace pds, in dependency order): the entry-point
// pipelines are templates with internal linkage, split by concern, not by compilation unit.
#pragma once

// ---------------------------------------------------------------------------------------------
// Host frames in row chunks.  A PDS_HOST frame of more than one chunk (256 MiB by default, PDS_HOST_CHUNK_MB) is never
// staged whole: its columns cross PCIe one row range at a time into ONE staging buffer of chunk size, the Gram kernel
// turns each range into an f64 moment record, and the records are summed in chunk order (fixed order: reproducible).
// HBM used: O(chunk) + (p+2)^2 doubles per chunk, whatever the frame size -- frames larger than HBM work, and the
// 13.6 GB headline frame needs 0.27 GB of staging instead of 13.6.  Everything is on the context's stream, so a chunk's
// copies wait for the previous chunk's kernel; the kernel is ~100x faster than the link, there is nothing to overlap.
// (reference: series_to_mat_for_lr copies the whole frame into one Vec, src/utils/mod.rs:101-206)
// ---------------------------------------------------------------------------------------------
static double env_mb(const char* name, double dflt) {
    const char* e = std::getenv(name);
    return e ? std::max(std::atof(e), 0.001) : dflt;
}
// frames up to g_host_resident_mb keep the whole-frame staging when a second pass over the rows follows (pred / resid):
// one trip over PCIe instead of two (default 96 GiB: a third of HBM).  pds_set_host_staging() changes both.
// Process-wide settings read by every calling thread (Polars' rayon threads each own a context) while pds_set_host_staging may
// write them: atomics, and an entry point takes ONE snapshot (StagingScope) that all of its sizing and staging decisions use, so
// the workspace bound it reserved and the buffers it then allocates cannot disagree.
std::atomic<double> g_host_chunk_mb{env_mb("PDS_HOST_CHUNK_MB", 256.0)};
std::atomic<double> g_host_resident_mb{env_mb("PDS_HOST_RESIDENT_MAX_MB", 98304.0)};
struct StagingSnap {
    size_t chunk_bytes, resident_bytes;
};
static thread_local const StagingSnap* t_staging = nullptr;
struct StagingScope {
    StagingSnap snap;
    const StagingSnap* prev;
    StagingScope() {
        snap.chunk_bytes = (size_t)(g_host_chunk_mb.load(std::memory_order_relaxed) * 1048576.0);
        snap.resident_bytes = (size_t)(g_host_resident_mb.load(std::memory_order_relaxed) * 1048576.0);
        prev = t_staging;
        if (!prev) t_staging = &snap;  // (the outermost scope of a call decides)
    }
    ~StagingScope() {
        if (!prev) t_staging = nullptr;
    }
};
static size_t host_chunk_bytes() {
    return t_staging ? t_staging->chunk_bytes : (size_t)(g_host_chunk_mb.load(std::memory_order_relaxed) * 1048576.0);
}
static size_t host_resident_max_bytes() {
    return t_staging ? t_staging->resident_bytes : (size_t)(g_host_resident_mb.load(std::memory_order_relaxed) * 1048576.0);
}
template <typename T>
static int64_t host_chunk_rows(int nc, int64_t n_rows) {
    int64_t r = (int64_t)(host_chunk_bytes() / ((size_t)nc * sizeof(T)));
    r = std::max<int64_t>(r & ~(int64_t)1023, 1024);  // whole 1 KiB pieces of every column
    return std::min(r, n_rows);
}
template <typename T>
static bool host_frame_is_chunked(int n_feat, bool weighted, int64_t n_rows) {
    const int nc = n_feat + 1 + (weighted ? 1 : 0);
    return n_feat <= kMaxFeatSmall && (size_t)n_rows * nc * sizeof(T) > host_chunk_bytes() && host_chunk_rows<T>(nc, n_rows) < n_rows;
}

struct ChunkStage {
    int nc = 0;
    int64_t rows = 0;       // rows per chunk
    size_t col_bytes = 0;   // bytes per staged column (256-byte aligned)
    char* base = nullptr;
};
// staging buffer for one chunk (+ `extra_cols` output columns of the same length) and its pointer table
template <typename T>
static int chunk_stage_init(pds_ctx* ctx, int n_feat, bool weighted, int64_t n_rows, int extra_cols, ChunkStage& st, DeviceCols<T>& dc) {
    st.nc = n_feat + 1 + (weighted ? 1 : 0);
    st.rows = host_chunk_rows<T>(st.nc, n_rows);
    st.col_bytes = ((size_t)st.rows * sizeof(T) + 255) & ~(size_t)255;
    if (int rc = ensure_ws(ctx, ctx->stage, st.col_bytes * (st.nc + extra_cols))) return rc;
    st.base = static_cast<char*>(ctx->stage.ptr);
    dc.nc = st.nc;
    dc.h_ptrs.resize(st.nc);
    for (int c = 0; c < st.nc; ++c) dc.h_ptrs[c] = reinterpret_cast<const T*>(st.base + st.col_bytes * c);
    dc.h_ptrs.resize(std::max(st.nc, 18), dc.h_ptrs[0]);
    dc.d_ptrs = reinterpret_cast<const T**>(ws_take(ctx, sizeof(T*) * dc.h_ptrs.size()));
    if (!dc.d_ptrs) return fail(PDS_ERR_HIP, "workspace allocation failed");
    PDS_HIP_CHECK(hipMemcpyAsync(dc.d_ptrs, dc.h_ptrs.data(), sizeof(T*) * dc.h_ptrs.size(), hipMemcpyHostToDevice, ctx->stream));
    return PDS_OK;
}
template <typename T>
static int chunk_stage_copy(pds_ctx* ctx, const ChunkStage& st, const T* const* cols /*[y, x1..xp]*/, const T* weights, int n_feat,
                            int64_t row0, int64_t rows) {
    for (int c = 0; c < st.nc; ++c) {
        const T* src = c < n_feat ? cols[c + 1] : (c == n_feat ? cols[0] : weights);  // device order x.., y, [w]
        PDS_HIP_CHECK(hipMemcpyAsync(st.base + st.col_bytes * c, src + row0, (size_t)rows * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    }
    return PDS_OK;
}
static size_t chunked_moments_workspace(int n_feat, int64_t n_rows, int64_t chunk_rows) {
    const size_t q = (size_t)n_feat + 2;
    return (size_t)((n_rows + chunk_rows - 1) / chunk_rows) * q * q * sizeof(double) + 4096 + 18 * sizeof(void*);
}

template <typename T>
static int moments_from_host_chunked(pds_ctx* ctx, const T* const* cols, const T* weights, int n_feat, int64_t n_rows, T* d_mom) {
    ChunkStage st;
    DeviceCols<T> dc;
    if (int rc = chunk_stage_init<T>(ctx, n_feat, weights != nullptr, n_rows, 0, st, dc)) return rc;
    const int q = n_feat + 2;
    const int nchunks = (int)((n_rows + st.rows - 1) / st.rows);
    double* d_slots = reinterpret_cast<double*>(ws_take(ctx, (size_t)nchunks * q * q * sizeof(double)));
    if (!d_slots) return fail(PDS_ERR_HIP, "workspace allocation failed");
    for (int k = 0; k < nchunks; ++k) {
        const int64_t row0 = (int64_t)k * st.rows, rows = std::min(st.rows, n_rows - row0);
        if (int rc = chunk_stage_copy<T>(ctx, st, cols, weights, n_feat, row0, rows)) return rc;
        if (int rc = launch_moments<T>(ctx, dc, n_feat, rows, weights != nullptr, nullptr, nullptr, 0, nullptr, d_slots + (size_t)k * q * q))
            return rc;
    }
    if (int rc = launch_sum_moment_slots<T>(ctx, d_slots, nchunks, q * q, d_mom)) return rc;
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (dc.h_ptrs: source of the table copy)
    return PDS_OK;
}

// second pass of pl_lr_pred over a host frame too large to keep in HBM: the rows cross PCIe again, chunk by chunk, and
// pred / resid go back as each chunk finishes
template <typename T>
static int pred_from_host_chunked(pds_ctx* ctx, const T* const* cols, int n_feat, int64_t n_rows, int add_bias, const T* d_coeffs,
                                  T* pred, T* resid) {
    ChunkStage st;
    DeviceCols<T> dc;
    if (int rc = chunk_stage_init<T>(ctx, n_feat, false, n_rows, 2, st, dc)) return rc;
    T* d_pred = reinterpret_cast<T*>(st.base + st.col_bytes * st.nc);
    T* d_resid = reinterpret_cast<T*>(st.base + st.col_bytes * (st.nc + 1));
    double* d_sums = reinterpret_cast<double*>(ws_take(ctx, 64));
    for (int64_t row0 = 0; row0 < n_rows; row0 += st.rows) {
        const int64_t rows = std::min(st.rows, n_rows - row0);
        if (int rc = chunk_stage_copy<T>(ctx, st, cols, (const T*)nullptr, n_feat, row0, rows)) return rc;
        if (int rc = launch_pass2<T>(ctx, dc, n_feat, rows, add_bias, false, d_coeffs, nullptr, 0, d_pred, d_resid, d_sums, nullptr)) return rc;
        if (pred) PDS_HIP_CHECK(hipMemcpyAsync(pred + row0, d_pred, (size_t)rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        if (resid) PDS_HIP_CHECK(hipMemcpyAsync(resid + row0, d_resid, (size_t)rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    }
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// Row-major matrices (the pyclass route: NumPy X of LR / ElasticNet / OnlineLR, numpy_faer.rs:10-66) -> column buffers in HBM
// ---------------------------------------------------------------------------------------------
template <typename T>
static int rows_to_cols_impl(pds_ctx* ctx, const T* X, int64_t ld, int64_t n_rows, int n_cols, pds_space space, T* out_cols,
                             int64_t col_stride) {
    if (!ctx || !X || !out_cols) return fail(PDS_ERR_INVALID, "null argument");
    if (n_rows <= 0 || n_cols <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    if (ld < n_cols || col_stride < n_rows) return fail(PDS_ERR_INVALID, "row stride < columns or column stride < rows");
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    if (space == PDS_DEVICE) {
        if (int rc = launch_rows_to_cols<T>(ctx, X, ld, n_rows, n_cols, out_cols, col_stride, 0)) return rc;
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return PDS_OK;
    }
    // host matrix: contiguous row chunks through one staging buffer (one copy per chunk; the kernel behind it is ~100x faster
    // than the link, so the single stream loses nothing)
    int64_t rows_per = (int64_t)(host_chunk_bytes() / ((size_t)ld * sizeof(T)));
    rows_per = std::min<int64_t>(std::max<int64_t>(rows_per & ~(int64_t)63, 64), n_rows);
    if (int rc = ensure_ws(ctx, ctx->stage, (size_t)rows_per * ld * sizeof(T) + 256)) return rc;
    T* d_stage = reinterpret_cast<T*>(ctx->stage.ptr);
    for (int64_t r0 = 0; r0 < n_rows; r0 += rows_per) {
        const int64_t rows = std::min(rows_per, n_rows - r0);
        // (the last row may be shorter than ld in the caller's allocation: copy rows - 1 full strides + n_cols values)
        const size_t bytes = ((size_t)(rows - 1) * ld + n_cols) * sizeof(T);
        PDS_HIP_CHECK(hipMemcpyAsync(d_stage, X + r0 * ld, bytes, hipMemcpyHostToDevice, ctx->stream));
        if (int rc = launch_rows_to_cols<T>(ctx, d_stage, ld, rows, n_cols, out_cols, col_stride, r0)) return rc;
    }
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}
