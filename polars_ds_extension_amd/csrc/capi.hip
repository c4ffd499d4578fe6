// capi.hip -- the extern "C" surface declared in include/pds_lstsq.h: context management, host <-> HBM
// staging, and the per-expression pipelines that string the kernels together.
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "common.hpp"

namespace pds {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

static hipEvent_t take_event(pds_ctx* ctx) {
    if (!ctx->ev_pool.empty()) {
        hipEvent_t e = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
KernelTimer::KernelTimer(pds_ctx* c, int k) : ctx(c), kind(k) {
    if (!ctx->timing) return;
    a = take_event(ctx);
    b = take_event(ctx);
    if (a) (void)hipEventRecord(a, ctx->stream);
}
KernelTimer::~KernelTimer() {
    if (!ctx->timing || !a || !b) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->ev_pending.push_back({kind, a, b});
}

int ensure_ws(pds_ctx* ctx, Workspace& w, size_t bytes) {
    if (bytes <= w.bytes) return PDS_OK;
    if (w.ptr) {
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        PDS_HIP_CHECK(hipFree(w.ptr));
        w.ptr = nullptr;
        w.bytes = 0;
    }
    size_t want = std::max(bytes, (size_t)1 << 20);
    want = (want + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    PDS_HIP_CHECK(hipMalloc(&w.ptr, want));
    w.bytes = want;
    return PDS_OK;
}

int ensure_pinned(pds_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->pinned_bytes) return PDS_OK;
    if (ctx->pinned) {
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        PDS_HIP_CHECK(hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
    }
    size_t want = std::max(bytes, (size_t)1 << 16);
    PDS_HIP_CHECK(hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
    ctx->pinned_bytes = want;
    return PDS_OK;
}

int ws_reserve(pds_ctx* ctx, size_t total_bytes) {
    ctx->ws_used = 0;
    for (void* q : ctx->ws_spill) (void)hipFree(q);  // (hipFree waits for the device: the previous call's kernels are done)
    ctx->ws_spill.clear();
    return ensure_ws(ctx, ctx->ws, total_bytes + 4096);
}
void* ws_take(pds_ctx* ctx, size_t bytes) {
    const size_t off = (ctx->ws_used + 255) & ~(size_t)255;
    if (off + bytes > ctx->ws.bytes) {
        // An entry point under-estimated its ws_reserve() bound.  Handing out memory past the workspace would corrupt
        // whatever lives behind it without any error, so the slice comes from its own allocation instead (released by the
        // next ws_reserve) and the event is counted: tests assert the counter stays at zero.
        void* q = nullptr;
        ++ctx->ws_spill_count;
        if (hipMalloc(&q, bytes + 256) != hipSuccess) return nullptr;  // (a null slice faults loudly in the kernel)
        ctx->ws_spill.push_back(q);
        return q;
    }
    ctx->ws_used = off + bytes;
    return static_cast<char*>(ctx->ws.ptr) + off;
}

constexpr size_t kSmallFrameBytes = (size_t)1 << 20;  // host frames up to this size are staged through pinned memory

template <typename T>
int make_device_cols(pds_ctx* ctx, const T* const* cols, const T* weights, int n_feat, int64_t n_rows,
                     pds_space space, DeviceCols<T>& out) {
    const int nc = n_feat + 1 + (weights ? 1 : 0);
    out.nc = nc;
    out.h_ptrs.resize(nc);
    std::vector<const T*> src(nc);
    for (int c = 0; c < n_feat; ++c) src[c] = cols[c + 1];  // reference order is [y, x1..xp]
    src[n_feat] = cols[0];
    if (weights) src[n_feat + 1] = weights;
    if (space == PDS_DEVICE) {
        for (int c = 0; c < nc; ++c) out.h_ptrs[c] = src[c];
    } else {
        const size_t col_bytes = ((size_t)n_rows * sizeof(T) + 255) & ~(size_t)255;
        const size_t tbl_entries = (size_t)std::max(nc, 18);
        const size_t tbl_bytes = (tbl_entries * sizeof(T*) + 255) & ~(size_t)255;
        if (tbl_bytes + col_bytes * nc <= kSmallFrameBytes) {
            // small frame (the per-group call pattern of Polars: ~100 rows): every pageable hipMemcpyAsync costs 5-8 us,
            // so gather the columns and the pointer table in pinned memory with the CPU and ship them in ONE copy
            const size_t total = tbl_bytes + col_bytes * nc;
            if (total > ctx->pinned_in_bytes) {
                if (ctx->pinned_in) {
                    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    PDS_HIP_CHECK(hipHostFree(ctx->pinned_in));
                    ctx->pinned_in = nullptr;
                    ctx->pinned_in_bytes = 0;
                }
                PDS_HIP_CHECK(hipHostMalloc(&ctx->pinned_in, kSmallFrameBytes, hipHostMallocDefault));
                ctx->pinned_in_bytes = kSmallFrameBytes;
            }
            if (int rc = ensure_ws(ctx, ctx->stage, total)) return rc;
            char* pin = static_cast<char*>(ctx->pinned_in);
            char* dev = static_cast<char*>(ctx->stage.ptr);
            for (int c = 0; c < nc; ++c) {
                std::memcpy(pin + tbl_bytes + col_bytes * c, src[c], (size_t)n_rows * sizeof(T));
                out.h_ptrs[c] = reinterpret_cast<const T*>(dev + tbl_bytes + col_bytes * c);
            }
            out.h_ptrs.resize(tbl_entries, out.h_ptrs[0]);
            std::memcpy(pin, out.h_ptrs.data(), tbl_entries * sizeof(T*));
            PDS_HIP_CHECK(hipMemcpyAsync(dev, pin, total, hipMemcpyHostToDevice, ctx->stream));
            out.d_ptrs = reinterpret_cast<const T**>(dev);
            // (the previous call's copy out of pinned_in has completed: every API call synchronises before returning)
            return PDS_OK;
        }
        // stage the host column buffers into HBM (one hipMemcpyAsync per column; see DESIGN.md for the
        // PCIe-inclusive rate -- the timed path of bench.py is device resident)
        if (int rc = ensure_ws(ctx, ctx->stage, col_bytes * nc)) return rc;
        for (int c = 0; c < nc; ++c) {
            T* dst = reinterpret_cast<T*>(static_cast<char*>(ctx->stage.ptr) + col_bytes * c);
            PDS_HIP_CHECK(hipMemcpyAsync(dst, src[c], (size_t)n_rows * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
            out.h_ptrs[c] = dst;
        }
    }
    // the device table always has 18 readable entries (16 features, y, w); unused ones alias column 0 so that
    // kernels may fetch the whole table with wide scalar loads
    out.h_ptrs.resize(std::max(nc, 18), out.h_ptrs[0]);
    out.d_ptrs = reinterpret_cast<const T**>(ws_take(ctx, sizeof(T*) * out.h_ptrs.size()));
    PDS_HIP_CHECK(hipMemcpyAsync(out.d_ptrs, out.h_ptrs.data(), sizeof(T*) * out.h_ptrs.size(), hipMemcpyHostToDevice, ctx->stream));
    // h_ptrs lives in `out` (caller's stack) until the call returns, and every API call synchronises
    // before returning, so the async copy source stays valid.
    return PDS_OK;
}

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
double g_host_chunk_mb = env_mb("PDS_HOST_CHUNK_MB", 256.0);
double g_host_resident_mb = env_mb("PDS_HOST_RESIDENT_MAX_MB", 98304.0);
static size_t host_chunk_bytes() { return (size_t)(g_host_chunk_mb * 1048576.0); }
static size_t host_resident_max_bytes() { return (size_t)(g_host_resident_mb * 1048576.0); }
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

// ---------------------------------------------------------------------------------------------
// host-side p' x p' SVD solve (solver = "svd", rcond path): one-sided Jacobi on the Gram matrix.
// O(p'^3) on a 2 KB matrix -- not worth a kernel; only reached for single systems.
// ---------------------------------------------------------------------------------------------
static bool jacobi_svd(const std::vector<double>& a, int n, std::vector<double>& u, std::vector<double>& s,
                       std::vector<double>& v) {
    u = a;
    v.assign((size_t)n * n, 0.0);
    s.assign(n, 0.0);
    for (int i = 0; i < n; ++i) v[i + (size_t)i * n] = 1.0;
    for (double x : u)
        if (!std::isfinite(x)) return false;
    const double eps = 2.220446049250313e-16;
    bool conv = false;
    for (int sweep = 0; sweep < 60 && !conv; ++sweep) {
        conv = true;
        for (int i = 0; i < n - 1; ++i)
            for (int j = i + 1; j < n; ++j) {
                double al = 0, be = 0, ga = 0;
                for (int r = 0; r < n; ++r) {
                    al += u[r + (size_t)i * n] * u[r + (size_t)i * n];
                    be += u[r + (size_t)j * n] * u[r + (size_t)j * n];
                    ga += u[r + (size_t)i * n] * u[r + (size_t)j * n];
                }
                if (ga == 0.0 || std::fabs(ga) <= eps * std::sqrt(al) * std::sqrt(be)) continue;  // (al * be may overflow)
                conv = false;
                const double zeta = (be - al) / (2 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
                const double c = 1 / std::sqrt(1 + t * t), sn = c * t;
                for (int r = 0; r < n; ++r) {
                    double x = u[r + (size_t)i * n], y = u[r + (size_t)j * n];
                    u[r + (size_t)i * n] = c * x - sn * y;
                    u[r + (size_t)j * n] = sn * x + c * y;
                    x = v[r + (size_t)i * n];
                    y = v[r + (size_t)j * n];
                    v[r + (size_t)i * n] = c * x - sn * y;
                    v[r + (size_t)j * n] = sn * x + c * y;
                }
            }
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) {
        double nn = 0;
        for (int r = 0; r < n; ++r) nn += u[r + (size_t)i * n] * u[r + (size_t)i * n];
        s[i] = std::sqrt(nn);
        if (s[i] > 0)
            for (int r = 0; r < n; ++r) u[r + (size_t)i * n] /= s[i];
        order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return s[x] > s[y]; });
    std::vector<double> u2(u.size()), v2(v.size()), s2(n);
    for (int k = 0; k < n; ++k) {
        s2[k] = s[order[k]];
        for (int r = 0; r < n; ++r) {
            u2[r + (size_t)k * n] = u[r + (size_t)order[k] * n];
            v2[r + (size_t)k * n] = v[r + (size_t)order[k] * n];
        }
    }
    u.swap(u2);
    v.swap(v2);
    s.swap(s2);
    return true;
}

// G (pp x pp) and rhs from a host copy of the moment matrix
template <typename T>
static void host_normal_eq(const std::vector<T>& M, int p, int bias, double lambda, std::vector<double>& G,
                           std::vector<double>& c) {
    const int pp = p + bias, q = p + 2;
    G.assign((size_t)pp * pp, 0.0);
    c.assign(pp, 0.0);
    for (int j = 0; j < pp; ++j) {
        for (int i = 0; i < pp; ++i) G[i + (size_t)j * pp] = (double)M[i + (size_t)j * q];
        c[j] = (double)M[j + (size_t)(p + 1) * q];
    }
    if (lambda > 0)
        for (int i = 0; i < p; ++i) G[i + (size_t)i * pp] += lambda;
}

struct Method {
    enum Kind { OLS, NNLS, CD } kind;
    double l1, l2;
    int positive;
};
static Method pick_method(const pds_lr_params* prm) {
    // LRMethods::from((l1, l2)) + the (method, positive) match of pl_lr: linear_regression.rs:447-497
    const bool l1 = prm->l1_reg > 0.0, l2 = prm->l2_reg > 0.0;
    Method m;
    m.positive = prm->positive ? 1 : 0;
    if (!l1) {  // Normal or L2
        if (!m.positive) return {Method::OLS, 0.0, prm->l2_reg, 0};
        if (!l2) return {Method::NNLS, 0.0, 0.0, 1};
        return {Method::CD, 0.0, prm->l2_reg, 1};
    }
    return {Method::CD, prm->l1_reg, l2 ? prm->l2_reg : 0.0, m.positive};
}

// Device moments -> coefficients on the host.  The f32 twin's iteration caps apply when T = float: coordinate descent 2000
// everywhere, NNLS 200 in `pl_lr_f32` (linear_regression_f32.rs:343) but 2000 in `pl_lr_pred_f32` (:620) -- `pred_path`.
// `force_cd`: ElasticNet::fit_unchecked (lr_solvers.rs:139-164) always runs faer_coordinate_descent, also for l1_reg <= 0.
template <typename T>
static int lr_from_device_moments(pds_ctx* ctx, const T* d_mom, int p, const pds_lr_params* prm, bool weighted,
                                  T* coeffs, int* is_null, T* d_coeffs_keep /*nullable device copy*/, bool pred_path = false,
                                  bool force_cd = false) {
    const int bias = prm->add_bias ? 1 : 0, pp = p + bias, q = p + 2;
    if (is_null) *is_null = 0;
    // coefficients and the null flag sit in one block so that they come back in one copy
    const size_t co_bytes = (sizeof(T) * (size_t)(pp + 2) + 15) & ~(size_t)15;
    char* d_blk = d_coeffs_keep ? nullptr : reinterpret_cast<char*>(ws_take(ctx, co_bytes + 16));
    T* d_coeffs = d_coeffs_keep ? d_coeffs_keep : reinterpret_cast<T*>(d_blk);
    uint8_t* d_flag = d_coeffs_keep ? reinterpret_cast<uint8_t*>(ws_take(ctx, 16)) : reinterpret_cast<uint8_t*>(d_blk + co_bytes);
    int* d_info = reinterpret_cast<int*>(ws_take(ctx, 16));
    if (int rc = ensure_pinned(ctx, 4096 + sizeof(T) * (size_t)(q * q + pp))) return rc;
    Method m = weighted ? Method{Method::OLS, 0.0, 0.0, 0} : pick_method(prm);
    if (force_cd) m = Method{Method::CD, prm->l1_reg > 0.0 ? prm->l1_reg : 0.0, prm->l2_reg > 0.0 ? prm->l2_reg : 0.0, prm->positive ? 1 : 0};
    const bool f32 = sizeof(T) == 4;
    if (m.kind == Method::OLS && prm->solver == PDS_SOLVER_SVD) {
        // svd: small host solve on the moments
        std::vector<T> M((size_t)q * q);
        PDS_HIP_CHECK(hipMemcpyAsync(M.data(), d_mom, sizeof(T) * M.size(), hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        std::vector<double> G, c, u, s, v;
        host_normal_eq(M, p, bias, weighted ? 0.0 : m.l2, G, c);
        const bool gate = !weighted && prm->singular_x_tol > 0.0;
        double ln_den = 0.0;
        bool null = false;
        if (gate)
            for (int i = 0; i < pp; ++i) {
                if (G[i + (size_t)i * pp] <= 0.0) null = true;
                else ln_den += std::log(G[i + (size_t)i * pp]);
            }
        bool ok = !null && jacobi_svd(G, pp, u, s, v);
        if (gate && !null) {
            if (!ok) null = true;  // "SVD failure -> treat as rank-deficient" lr_solvers.rs:361-362
            else {
                double ln_det = 0.0;
                for (int i = 0; i < pp; ++i) ln_det += std::log(s[i]);
                if (ln_det - ln_den <= std::log(prm->singular_x_tol)) null = true;
            }
        }
        if (null) {
            for (int i = 0; i < pp; ++i) coeffs[i] = (T)NAN;
            if (is_null) *is_null = 1;
        } else if (ok) {
            std::vector<double> z(pp);
            for (int i = 0; i < pp; ++i) {
                double acc = 0;
                for (int r = 0; r < pp; ++r) acc += u[r + (size_t)i * pp] * c[r];
                z[i] = acc / s[i];
            }
            for (int r = 0; r < pp; ++r) {
                double acc = 0;
                for (int i = 0; i < pp; ++i) acc += v[r + (size_t)i * pp] * z[i];
                coeffs[r] = (T)acc;
            }
        } else {
            // ungated SVD failure falls back to QR (lr_solvers.rs:284-287)
            SolveParams sp{p, bias, PDS_SOLVER_QR, m.l2, 0.0, 0};
            if (int rc = launch_solve<T>(ctx, d_mom, 1, sp, d_coeffs, d_flag, nullptr, nullptr)) return rc;
            PDS_HIP_CHECK(hipMemcpyAsync(coeffs, d_coeffs, sizeof(T) * pp, hipMemcpyDeviceToHost, ctx->stream));
            PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
        if (d_coeffs_keep)
            PDS_HIP_CHECK(hipMemcpyAsync(d_coeffs_keep, coeffs, sizeof(T) * pp, hipMemcpyHostToDevice, ctx->stream));
        return PDS_OK;
    }
    if (m.kind == Method::OLS) {
        SolveParams sp{p, bias, prm->solver, weighted ? 0.0 : m.l2, weighted ? 0.0 : prm->singular_x_tol, 0};
        if (int rc = launch_solve<T>(ctx, d_mom, 1, sp, d_coeffs, d_flag, nullptr, nullptr)) return rc;
    } else if (m.kind == Method::NNLS) {
        if (int rc = launch_nnls<T>(ctx, d_mom, p, bias, prm->tol, f32 ? (pred_path ? 2000 : 200) : prm->max_iter, d_coeffs)) return rc;
        PDS_HIP_CHECK(hipMemsetAsync(d_flag, 0, 1, ctx->stream));
    } else {
        if (int rc = launch_cd<T>(ctx, d_mom, p, bias, m.l1, m.l2, prm->tol, (f32 && !force_cd) ? 2000 : prm->max_iter, m.positive,
                                  d_coeffs, d_info))
            return rc;
        PDS_HIP_CHECK(hipMemsetAsync(d_flag, 0, 1, ctx->stream));
    }
    char* pin = static_cast<char*>(ctx->pinned);
    if (d_blk && co_bytes + 16 <= 2048) {
        PDS_HIP_CHECK(hipMemcpyAsync(pin, d_blk, co_bytes + 16, hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        std::memcpy(coeffs, pin, sizeof(T) * pp);
        if (is_null) *is_null = pin[co_bytes] ? 1 : 0;
        return PDS_OK;
    }
    PDS_HIP_CHECK(hipMemcpyAsync(pin, d_coeffs, sizeof(T) * pp, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipMemcpyAsync(pin + sizeof(T) * (size_t)(pp + 2) + 64, d_flag, 1, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    std::memcpy(coeffs, pin, sizeof(T) * pp);
    if (is_null) *is_null = pin[sizeof(T) * (size_t)(pp + 2) + 64] ? 1 : 0;
    return PDS_OK;
}

static int check_shape(int n_feat, int64_t n_rows, int add_bias) {
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_rows == 0) return fail(PDS_ERR_EMPTY, "Empty data");  // linear_regression.rs:166-168
    if (n_rows < n_feat + (add_bias ? 1 : 0))
        return fail(PDS_ERR_TOO_FEW_ROWS, "#Data < #features. No conclusive result.");  // :169-173
    return PDS_OK;
}

template <typename T>
static int lr_impl(pds_ctx* ctx, const T* const* cols, const T* weights, int n_feat, int64_t n_rows, pds_space space,
                   const pds_lr_params* prm, T* coeffs, int* is_null, T* pred, T* resid, bool force_cd = false) {
    if (!ctx || !cols || !prm || !coeffs) return fail(PDS_ERR_INVALID, "null argument");
    if (force_cd) {  // ElasticNet::fit (lr/mod.rs:114-125) only rejects an empty frame; fewer rows than columns is fine
        if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
        if (n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    } else if (int rc = check_shape(n_feat, n_rows, prm->add_bias)) {
        return rc;
    }
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int q = n_feat + 2, pp = n_feat + (prm->add_bias ? 1 : 0);
    const bool want_pred = pred || resid;
    // host frames of more than one chunk are streamed through a chunk-sized staging buffer; with a residual pass behind the
    // fit only when keeping the frame in HBM is not an option (then the rows cross PCIe twice)
    const int nc_host = n_feat + 1 + (weights ? 1 : 0);
    bool chunked = space == PDS_HOST && host_frame_is_chunked<T>(n_feat, weights != nullptr, n_rows);
    if (chunked && want_pred && ((size_t)n_rows * nc_host * sizeof(T) <= host_resident_max_bytes() || weights)) chunked = false;
    size_t need = 65536 + sizeof(T) * (size_t)q * q + sizeof(T*) * (size_t)(n_feat + 32);
    if (n_feat > kMaxFeatSmall) need += moments_wide_workspace(ctx->num_cus, n_feat, n_rows, weights != nullptr);
    if (want_pred && space == PDS_HOST && !chunked) need += 2 * ((size_t)n_rows * sizeof(T) + 512);
    if (chunked) need += 2 * chunked_moments_workspace(n_feat, n_rows, host_chunk_rows<T>(nc_host, n_rows));
    if (int rc = ws_reserve(ctx, need)) return rc;
    DeviceCols<T> dc;
    T* d_mom = nullptr;
    if (chunked) {
        d_mom = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * q * q));
        if (int rc = moments_from_host_chunked<T>(ctx, cols, weights, n_feat, n_rows, d_mom)) return rc;
    } else {
        if (int rc = make_device_cols<T>(ctx, cols, weights, n_feat, n_rows, space, dc)) return rc;
        d_mom = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * q * q));
        if (int rc = launch_moments<T>(ctx, dc, n_feat, n_rows, weights != nullptr, d_mom)) return rc;
    }
    T* d_coeffs = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * (pp + 2)));
    int null_flag = 0;
    // (a device copy of the coefficients is only kept for the residual pass: without it they come back with the null
    //  flag in one copy)
    if (int rc = lr_from_device_moments<T>(ctx, d_mom, n_feat, prm, weights != nullptr, coeffs, &null_flag, want_pred ? d_coeffs : nullptr,
                                           want_pred, force_cd))
        return rc;
    if (is_null) *is_null = null_flag;
    if (want_pred && chunked) return pred_from_host_chunked<T>(ctx, cols, n_feat, n_rows, prm->add_bias, d_coeffs, pred, resid);
    if (want_pred) {
        T* d_pred = pred;
        T* d_resid = resid;
        if (space == PDS_HOST) {
            d_pred = reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T)));
            d_resid = reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T)));
        }
        double* d_sums = reinterpret_cast<double*>(ws_take(ctx, 64));
        // a gated fit yields all-null pred/resid in the reference (:745-750); here NaN coefficients
        // propagate to NaN rows and the caller marks them invalid through *is_null.
        if (int rc = launch_pass2<T>(ctx, dc, n_feat, n_rows, prm->add_bias, false, d_coeffs, nullptr, 0, d_pred, d_resid,
                                     d_sums, nullptr))
            return rc;
        if (space == PDS_HOST) {
            if (pred) PDS_HIP_CHECK(hipMemcpyAsync(pred, d_pred, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
            if (resid) PDS_HIP_CHECK(hipMemcpyAsync(resid, d_resid, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// pl_lr / pl_lr_pred with Arrow validity bitmaps: null policy on the device, then the ordinary fit
// ---------------------------------------------------------------------------------------------
template <typename T>
static int lr_nullable_impl(pds_ctx* ctx, const T* const* cols, const uint8_t* const* validity, const int64_t* bit_offsets,
                            int n_feat, int64_t n_rows, pds_space space, int policy, T fill_value,
                            const pds_lr_params* prm, T* coeffs, int* is_null, T* pred, T* resid, uint8_t* row_valid,
                            int64_t* n_used) {
    if (!ctx || !cols || !prm || !coeffs) return fail(PDS_ERR_INVALID, "null argument");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_rows == 0) return fail(PDS_ERR_EMPTY, "Empty data");
    if (policy < PDS_NULL_RAISE || policy > PDS_NULL_IGNORE) return fail(PDS_ERR_INVALID, "Invalid NullPolicy.");
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int nc = n_feat + 1, q = n_feat + 2, pp = n_feat + (prm->add_bias ? 1 : 0);
    const bool want_pred = pred || resid;
    size_t need = (1 << 20) + sizeof(T) * (size_t)q * q + null_policy_workspace(nc, n_rows, sizeof(T));
    if (n_feat > kMaxFeatSmall) need += moments_wide_workspace(ctx->num_cus, n_feat, n_rows);
    if (space == PDS_HOST) need += (size_t)nc * ((size_t)n_rows / 8 + 4096);
    if (want_pred) need += 4 * ((size_t)n_rows * sizeof(T) + 512) + (size_t)n_rows + 512;
    if (int rc = ws_reserve(ctx, need)) return rc;
    DeviceCols<T> dc;
    if (int rc = make_device_cols<T>(ctx, cols, (const T*)nullptr, n_feat, n_rows, space, dc)) return rc;
    std::vector<const T*> ref_order(nc);
    ref_order[0] = dc.h_ptrs[n_feat];
    for (int c = 0; c < n_feat; ++c) ref_order[c + 1] = dc.h_ptrs[c];
    std::vector<const uint8_t*> bms(nc, nullptr);
    std::vector<int64_t> boff(nc, 0);
    for (int c = 0; c < nc; ++c) {
        boff[c] = bit_offsets ? bit_offsets[c] : 0;
        const uint8_t* b = validity ? validity[c] : nullptr;
        if (b && space == PDS_HOST) {
            const size_t bytes = (size_t)((boff[c] + n_rows + 7) / 8);
            uint8_t* d = reinterpret_cast<uint8_t*>(ws_take(ctx, bytes));
            PDS_HIP_CHECK(hipMemcpyAsync(d, b, bytes, hipMemcpyHostToDevice, ctx->stream));
            b = d;
        }
        bms[c] = b;
    }
    NullPrepared<T> prep;
    if (int rc = apply_null_policy<T>(ctx, ref_order, bms, boff, n_rows, policy, fill_value, prep)) return rc;
    if (n_used) *n_used = prep.n_kept;
    if (prep.n_kept == 0) return fail(PDS_ERR_EMPTY, "Empty data");
    if (prep.n_kept < pp) return fail(PDS_ERR_TOO_FEW_ROWS, "#Data < #features. No conclusive result.");
    DeviceCols<T> dk;
    dk.nc = nc;
    dk.h_ptrs.resize(nc);
    for (int c = 0; c < n_feat; ++c) dk.h_ptrs[c] = prep.cols[c + 1];
    dk.h_ptrs[n_feat] = prep.cols[0];
    dk.h_ptrs.resize(std::max(nc, 18), dk.h_ptrs[0]);
    dk.d_ptrs = reinterpret_cast<const T**>(ws_take(ctx, sizeof(T*) * dk.h_ptrs.size()));
    PDS_HIP_CHECK(hipMemcpyAsync(dk.d_ptrs, dk.h_ptrs.data(), sizeof(T*) * dk.h_ptrs.size(), hipMemcpyHostToDevice, ctx->stream));
    T* d_mom = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * q * q));
    if (int rc = launch_moments<T>(ctx, dk, n_feat, prep.n_kept, false, d_mom)) return rc;
    T* d_coeffs = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * (pp + 2)));
    int null_flag = 0;
    if (int rc = lr_from_device_moments<T>(ctx, d_mom, n_feat, prm, false, coeffs, &null_flag, want_pred ? d_coeffs : nullptr, want_pred)) return rc;
    if (is_null) *is_null = null_flag;
    if (want_pred) {
        T* c_pred = reinterpret_cast<T*>(ws_take(ctx, (size_t)prep.n_kept * sizeof(T)));
        T* c_resid = reinterpret_cast<T*>(ws_take(ctx, (size_t)prep.n_kept * sizeof(T)));
        double* d_sums = reinterpret_cast<double*>(ws_take(ctx, 64));
        if (int rc = launch_pass2<T>(ctx, dk, n_feat, prep.n_kept, prm->add_bias, false, d_coeffs, nullptr, 0, c_pred, c_resid,
                                     d_sums, nullptr))
            return rc;
        T* o_pred = pred;
        T* o_resid = resid;
        uint8_t* o_valid = row_valid;
        if (space == PDS_HOST) {
            o_pred = reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T)));
            o_resid = reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T)));
            o_valid = reinterpret_cast<uint8_t*>(ws_take(ctx, (size_t)n_rows));
        }
        if (prep.dropped) {
            if (o_pred) if (int rc = expand_rows<T>(ctx, c_pred, prep.d_keep, prep.d_rank, n_rows, o_pred, o_valid)) return rc;
            if (o_resid) if (int rc = expand_rows<T>(ctx, c_resid, prep.d_keep, prep.d_rank, n_rows, o_resid, nullptr)) return rc;
        } else {
            if (o_pred) PDS_HIP_CHECK(hipMemcpyAsync(o_pred, c_pred, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
            if (o_resid) PDS_HIP_CHECK(hipMemcpyAsync(o_resid, c_resid, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
            if (o_valid) PDS_HIP_CHECK(hipMemsetAsync(o_valid, 1, (size_t)n_rows, ctx->stream));
        }
        if (space == PDS_HOST) {
            if (pred) PDS_HIP_CHECK(hipMemcpyAsync(pred, o_pred, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
            if (resid) PDS_HIP_CHECK(hipMemcpyAsync(resid, o_resid, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
            if (row_valid) PDS_HIP_CHECK(hipMemcpyAsync(row_valid, o_valid, (size_t)n_rows, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// pl_lr_multi / pl_lr_multi_pred: k targets, one Gram build (targets 1..k-1 ride along as Gram columns)
// ---------------------------------------------------------------------------------------------
template <typename T>
static int lr_multi_impl(pds_ctx* ctx, const T* const* cols, int k, int n_feat, int64_t n_rows, pds_space space,
                         int add_bias, double l2_reg, int solver, double gate_tol, T* coeffs, int* is_null, T* pred,
                         T* resid) {
    if (!ctx || !cols || !coeffs) return fail(PDS_ERR_INVALID, "null argument");
    if (k < 1) return fail(PDS_ERR_INVALID, "need at least one target");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_rows == 0) return fail(PDS_ERR_EMPTY, "Empty data");  // series_to_mat_for_multi_lr :285-288
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int p = n_feat, bias = add_bias ? 1 : 0, pp = p + bias, q = p + 2;
    const int pa = p + k - 1, qa = pa + 2;  // augmented feature count: [x.., t_1..t_{k-1}], target t_0
    const bool want_pred = pred || resid;
    size_t need = (1 << 20) + sizeof(T) * ((size_t)qa * qa + (size_t)k * (q * q + pp + 2)) + sizeof(T*) * (size_t)(pa + 64) +
                  (size_t)k * (sizeof(T*) * (size_t)(p + 20) + 256);
    if (pa > kMaxFeatSmall) need += moments_wide_workspace(ctx->num_cus, pa, n_rows);
    if (want_pred && space == PDS_HOST) need += 2 * ((size_t)n_rows * sizeof(T) + 512);
    if (int rc = ws_reserve(ctx, need)) return rc;
    // reference order for make_device_cols is [y, x1..]: y = t_0, features = x_1..x_p, t_1..t_{k-1}
    std::vector<const T*> order(pa + 1);
    order[0] = cols[0];
    for (int c = 0; c < p; ++c) order[1 + c] = cols[k + c];
    for (int i = 1; i < k; ++i) order[p + i] = cols[i];
    DeviceCols<T> dc;
    if (int rc = make_device_cols<T>(ctx, order.data(), (const T*)nullptr, pa, n_rows, space, dc)) return rc;
    T* d_moma = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * qa * qa));
    if (int rc = launch_moments<T>(ctx, dc, pa, n_rows, false, d_moma)) return rc;
    std::vector<T> Ma((size_t)qa * qa);
    PDS_HIP_CHECK(hipMemcpyAsync(Ma.data(), d_moma, sizeof(T) * Ma.size(), hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // per-target (p+2)^2 moment matrices: same X'X / column sums / n, its own X't, sum t, t't
    std::vector<T> Mk((size_t)k * q * q, T(0));
    auto A = [&](int i, int j) { return Ma[i + (size_t)j * qa]; };
    for (int t = 0; t < k; ++t) {
        T* M = Mk.data() + (size_t)t * q * q;
        const int ti = (t == 0) ? pa + 1 : p + t - 1;  // index of target t inside the augmented matrix
        for (int j = 0; j < p; ++j) {
            for (int i = 0; i < p; ++i) M[i + j * q] = A(i, j);
            M[j + p * q] = M[p + j * q] = A(j, pa);           // column sums
            M[j + (p + 1) * q] = M[(p + 1) + j * q] = A(j, ti);  // X't
        }
        M[p + p * q] = A(pa, pa);                               // n
        M[p + (p + 1) * q] = M[(p + 1) + p * q] = A(pa, ti);    // sum t
        M[(p + 1) + (p + 1) * q] = A(ti, ti);
    }
    T* d_mk = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * Mk.size()));
    T* d_co = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * (size_t)k * pp));
    uint8_t* d_fl = reinterpret_cast<uint8_t*>(ws_take(ctx, (size_t)k + 16));
    PDS_HIP_CHECK(hipMemcpyAsync(d_mk, Mk.data(), sizeof(T) * Mk.size(), hipMemcpyHostToDevice, ctx->stream));
    SolveParams sp{p, bias, solver == PDS_SOLVER_SVD ? PDS_SOLVER_QR : solver, l2_reg, gate_tol, 0};
    if (int rc = launch_solve<T>(ctx, d_mk, k, sp, d_co, d_fl, nullptr, nullptr)) return rc;
    std::vector<uint8_t> fl(k);
    PDS_HIP_CHECK(hipMemcpyAsync(coeffs, d_co, sizeof(T) * (size_t)k * pp, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipMemcpyAsync(fl.data(), d_fl, (size_t)k, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (is_null) *is_null = fl[0] ? 1 : 0;
    if (want_pred) {
        const int tl = std::max(18, p + 2);  // pointer table length (the p <= 16 kernels read 18 entries)
        double* d_sums = reinterpret_cast<double*>(ws_take(ctx, 64));
        T* t_pred = nullptr;
        T* t_resid = nullptr;
        if (space == PDS_HOST) {
            t_pred = reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T)));
            t_resid = reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T)));
        }
        for (int t = 0; t < k; ++t) {
            DeviceCols<T> dt;
            dt.nc = p + 1;
            dt.h_ptrs.assign(dc.h_ptrs.begin(), dc.h_ptrs.begin() + p);
            dt.h_ptrs.push_back(t == 0 ? dc.h_ptrs[pa] : dc.h_ptrs[p + t - 1]);
            dt.h_ptrs.resize(tl, dt.h_ptrs[0]);
            dt.d_ptrs = reinterpret_cast<const T**>(ws_take(ctx, sizeof(T*) * tl));
            PDS_HIP_CHECK(hipMemcpyAsync(dt.d_ptrs, dt.h_ptrs.data(), sizeof(T*) * tl, hipMemcpyHostToDevice, ctx->stream));
            T* op = (space == PDS_HOST) ? t_pred : (pred ? pred + (size_t)t * n_rows : nullptr);
            T* orr = (space == PDS_HOST) ? t_resid : (resid ? resid + (size_t)t * n_rows : nullptr);
            if (int rc = launch_pass2<T>(ctx, dt, p, n_rows, bias, false, d_co + (size_t)t * pp, nullptr, 0, op, orr, d_sums, nullptr))
                return rc;
            if (space == PDS_HOST) {
                if (pred) PDS_HIP_CHECK(hipMemcpyAsync(pred + (size_t)t * n_rows, t_pred, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
                if (resid) PDS_HIP_CHECK(hipMemcpyAsync(resid + (size_t)t * n_rows, t_resid, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
            }
            PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // dt.h_ptrs goes out of scope
        }
    }
    return PDS_OK;
}

template <typename T>
static int moments_impl(pds_ctx* ctx, const T* const* cols, const T* weights, int n_feat, int64_t n_rows,
                        pds_space space, T* moments, pds_space out_space) {
    if (!ctx || !cols || !moments) return fail(PDS_ERR_INVALID, "null argument");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int q = n_feat + 2;
    size_t need = 65536 + sizeof(T) * (size_t)q * q + sizeof(T*) * (size_t)(n_feat + 32) +
                  (n_feat > kMaxFeatSmall ? moments_wide_workspace(ctx->num_cus, n_feat, n_rows, weights != nullptr) : 0);
    if (space == PDS_HOST && host_frame_is_chunked<T>(n_feat, weights != nullptr, n_rows))
        need += 2 * chunked_moments_workspace(n_feat, n_rows, host_chunk_rows<T>(n_feat + 1 + (weights ? 1 : 0), n_rows));
    if (int rc = ws_reserve(ctx, need)) return rc;
    DeviceCols<T> dc;
    T* d_mom = out_space == PDS_DEVICE ? moments : reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * q * q));
    if (space == PDS_HOST && host_frame_is_chunked<T>(n_feat, weights != nullptr, n_rows)) {
        if (int rc = moments_from_host_chunked<T>(ctx, cols, weights, n_feat, n_rows, d_mom)) return rc;
    } else {
        if (int rc = make_device_cols<T>(ctx, cols, weights, n_feat, n_rows, space, dc)) return rc;
        if (int rc = launch_moments<T>(ctx, dc, n_feat, n_rows, weights != nullptr, d_mom)) return rc;
    }
    if (out_space == PDS_HOST) {
        PDS_HIP_CHECK(hipMemcpyAsync(moments, d_mom, sizeof(T) * q * q, hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    } else if (space == PDS_HOST) {
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // staging buffers / pointer array must outlive the kernel
    } else {
        // device in, device out: the pointer array was copied from dc.h_ptrs (stack) -> wait for that copy only
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return PDS_OK;
}

template <typename T>
static int from_moments_impl(pds_ctx* ctx, const T* moments, pds_space mom_space, int n_feat, const pds_lr_params* prm,
                             T* coeffs, int* is_null) {
    if (!ctx || !moments || !prm || !coeffs) return fail(PDS_ERR_INVALID, "null argument");
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int q = n_feat + 2;
    if (int rc = ws_reserve(ctx, 65536 + sizeof(T) * (size_t)q * q)) return rc;
    const T* d_mom = moments;
    if (mom_space == PDS_HOST) {
        T* tmp = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * q * q));
        PDS_HIP_CHECK(hipMemcpyAsync(tmp, moments, sizeof(T) * q * q, hipMemcpyHostToDevice, ctx->stream));
        d_mom = tmp;
    }
    int rc = lr_from_device_moments<T>(ctx, d_mom, n_feat, prm, false, coeffs, is_null, nullptr);
    if (rc) return rc;
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// pl_lr_w_rcond(_f32) -> faer_solve_lr_rcond (lr_solvers.rs:216-258): SVD of X'X (+ lambda), singular values of X =
// sqrt of its eigenvalues, pseudo-inverse with the reference's cut-off rule (eigenvalue compared with rcond * s_max, as
// written at :226-240).  The Gram build is the device pass; the p' x p' decomposition is a host Jacobi SVD in f64 for
// both precisions (the f32 twin's moments are f32 -- what its matrix-core tiles produce -- the decomposition of the 2 KB
// matrix is not where its error comes from).
// ---------------------------------------------------------------------------------------------
template <typename T>
static int lr_rcond_impl(pds_ctx* ctx, const T* const* cols, int n_feat, int64_t n_rows, pds_space space, int add_bias,
                         double l2_reg, double rcond, T* coeffs, T* singular_values) {
    if (!ctx || !cols || !coeffs || !singular_values) return fail(PDS_ERR_INVALID, "null argument");
    if (int rc = check_shape(n_feat, n_rows, add_bias)) return rc;
    const int bias = add_bias ? 1 : 0, pp = n_feat + bias, q = n_feat + 2;
    std::vector<T> M((size_t)q * q);
    if (int rc = moments_impl<T>(ctx, cols, nullptr, n_feat, n_rows, space, M.data(), PDS_HOST)) return rc;
    std::vector<double> G, c, u, s, v;
    host_normal_eq(M, n_feat, bias, l2_reg, G, c);
    if (!jacobi_svd(G, pp, u, s, v)) return fail(PDS_ERR_NUMERIC, "SVD failed.");
    std::vector<double> sv(pp);
    for (int i = 0; i < pp; ++i) {
        sv[i] = std::sqrt(s[i]);
        singular_values[i] = (T)sv[i];
    }
    const double thr = rcond * sv[0];  // lr_solvers.rs:230-240 (eigenvalue vs rcond * s_max, as written)
    std::vector<double> z(pp);
    for (int i = 0; i < pp; ++i) {
        const double sinv = s[i] >= thr ? 1.0 / s[i] : 0.0;
        double acc = 0;
        for (int r = 0; r < pp; ++r) acc += u[r + (size_t)i * pp] * c[r];
        z[i] = acc * sinv;
    }
    for (int r = 0; r < pp; ++r) {
        double acc = 0;
        for (int i = 0; i < pp; ++i) acc += v[r + (size_t)i * pp] * z[i];
        coeffs[r] = (T)acc;
    }
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// GLM by iteratively re-weighted least squares: the caller of faer_weighted_lr (faer_irls, glm_solvers.rs:249-368;
// GLM::fit_unchecked :216-240).  One IRLS iteration = ONE pass over the frame (moments.hip WM = 3 forms the weights and the
// working response from the previous coefficients while the row is in registers) + a p' x p' pivoted-QR solve; the state
// between iterations is the coefficient vector, not four n-long vectors.
// ---------------------------------------------------------------------------------------------
template <typename T>
static int glm_irls_impl(pds_ctx* ctx, const T* const* cols, int n_feat, int64_t n_rows, pds_space space, int add_bias, int link,
                         int variance, T tol, int max_iter, T* coeffs, int* n_iter) {
    if (!ctx || !cols || !coeffs) return fail(PDS_ERR_INVALID, "null argument");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_feat > kMaxFeatSmall) return fail(PDS_ERR_UNSUPPORTED, "GLM (IRLS): up to 16 feature columns");
    if (n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    if (max_iter < 1) return fail(PDS_ERR_INVALID, "`max_iter` must be > 1.");  // linear_models.py:756-757
    if (link < 0 || link > 3 || variance < 0 || variance > 3) return fail(PDS_ERR_INVALID, "unknown link / variance function");
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int p = n_feat, bias = add_bias ? 1 : 0, pp = p + bias, q = p + 2;
    if (int rc = ws_reserve(ctx, 262144 + (size_t)max_iter * 1024 + sizeof(T) * (size_t)(2 * q * q + 2 * pp + 16) + sizeof(T*) * 64)) return rc;
    DeviceCols<T> dc;
    if (int rc = make_device_cols<T>(ctx, cols, (const T*)nullptr, p, n_rows, space, dc)) return rc;
    T* d_mom = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * q * q));
    T* d_beta = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * (pp + 2)));
    IrlsArgs ia;
    ia.link = link;
    ia.variance = variance;
    ia.init = 1;
    if (variance != 2) {  // mean of y for the starting mu (:272-279): sum(y) is an entry of the plain moment matrix
        if (int rc = launch_moments<T>(ctx, dc, p, n_rows, false, d_mom)) return rc;
        T sy = T(0);
        PDS_HIP_CHECK(hipMemcpyAsync(&sy, d_mom + p + (size_t)(p + 1) * q, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ia.y_mean = (double)sy / (double)n_rows;
    }
    pds_lr_params prm{};
    prm.add_bias = bias;
    prm.solver = PDS_SOLVER_QR;  // GLM::fit_unchecked passes LRSolverMethods::QR (:226, :236)
    prm.max_iter = 1;
    std::vector<T> beta(pp, T(0)), bnew(pp, T(0));
    int it = 0;
    while (it < max_iter) {
        ++it;
        if (int rc = launch_moments<T>(ctx, dc, p, n_rows, false, d_mom, d_beta, bias, nullptr, nullptr, &ia)) return rc;
        int null_flag = 0;
        if (int rc = lr_from_device_moments<T>(ctx, d_mom, p, &prm, /*weighted=*/true, bnew.data(), &null_flag, d_beta)) return rc;
        ia.init = 0;
        T max_diff = T(0);
        for (int j = 0; j < pp; ++j) max_diff = std::max(max_diff, (T)std::fabs(beta[j] - bnew[j]));
        beta = bnew;
        if (max_diff < tol) break;  // (a NaN difference never converges, as in the reference: :339-350)
    }
    for (int j = 0; j < pp; ++j) coeffs[j] = beta[j];
    if (n_iter) *n_iter = it;
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// lin_reg_report / wls_report
// ---------------------------------------------------------------------------------------------
// second pass over the frame: residuals (sum e^2, sum w e^2) and, for the HC estimators, the per-row weights s_i followed by
// one more *weighted* Gram build = the meat X' diag(s) X (d_mom2: (p+2)^2 moment layout; untouched for plain standard errors)
template <typename T>
static int report_second_pass(pds_ctx* ctx, const DeviceCols<T>& dc, int p, int64_t n_rows, int bias, bool weighted, int se_type,
                              const T* d_beta, const T* d_inv, double* d_sums, T* d_mom2) {
    const int hc = (se_type == PDS_SE) ? 0 : (se_type == PDS_HC2 ? 2 : (se_type == PDS_HC3 ? 3 : 1));
    // HC0 / HC1, p <= 16: the row weights are e_i^2, which the Gram kernel can form itself from the row it has just loaded --
    // residuals, sum e^2 and the meat in ONE pass over the frame (the report is two streams, not three, and the n-row
    // weight vector never exists).  HC2 / HC3 need the leverages (O(p'^2) per row): pass2_kernel + a weighted Gram build.
    static const bool no_fuse = [] { const char* e = std::getenv("PDS_REPORT_NO_FUSE"); return e && e[0] == '1'; }();
    if (hc == 1 && !weighted && p <= kMaxFeatSmall && !no_fuse)
        return launch_moments<T>(ctx, dc, p, n_rows, false, d_mom2, d_beta, bias, d_sums);
    T* d_s = hc ? reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T))) : nullptr;
    if (int rc = launch_pass2<T>(ctx, dc, p, n_rows, bias, weighted, d_beta, d_inv, hc, nullptr, nullptr, d_sums,
                                 reinterpret_cast<double*>(d_s)))
        return rc;
    if (hc) {
        // meat = X' diag(s) X : one more weighted Gram build with w = s
        DeviceCols<T> dc2;
        dc2.nc = p + 2;
        dc2.h_ptrs.assign(dc.h_ptrs.begin(), dc.h_ptrs.begin() + p + 1);
        dc2.h_ptrs.push_back(d_s);
        dc2.h_ptrs.resize(std::max(p + 2, 18), dc2.h_ptrs[0]);  // (the p <= 16 kernels fetch 18 entries with wide loads)
        dc2.d_ptrs = reinterpret_cast<const T**>(ws_take(ctx, sizeof(T*) * dc2.h_ptrs.size()));
        PDS_HIP_CHECK(hipMemcpyAsync(dc2.d_ptrs, dc2.h_ptrs.data(), sizeof(T*) * dc2.h_ptrs.size(), hipMemcpyHostToDevice, ctx->stream));
        if (int rc = launch_moments<T>(ctx, dc2, p, n_rows, true, d_mom2)) return rc;
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (dc2.h_ptrs is the source of the async table copy)
    }
    return PDS_OK;
}

// ---- O(p'^2) host epilogue (linear_regression.rs:861-939): r2 / adj_r2, standard errors, t, p, confidence interval
template <typename T, typename R>
static void report_epilogue(int64_t n_rows, int p, int bias, int se_type, bool weighted, T y_var, const T* beta, const T* inv,
                            const T* meat /*(p+2)^2 moment layout, HC only*/, const double* sums, R* out) {
    const int pp = p + bias, q = p + 2;
    const T dof = (T)n_rows - (T)pp;
    const T nf = (T)n_rows;
    const T ssr = (T)sums[0];
    const T ratio = ssr / (y_var * nf);
    out->r2 = (T)1 - ratio;
    out->adj_r2 = (T)1 - ratio * (((T)(n_rows - 1)) / (dof - (T)1));
    std::vector<T> se(pp);
    if (se_type == PDS_SE) {
        const T mse = (weighted ? (T)sums[1] : ssr) / dof;
        for (int i = 0; i < pp; ++i) se[i] = (T)std::sqrt((double)(mse * inv[i + (size_t)i * pp]));
    } else {
        // var_hc_ii = inv_i . meat . inv_i ; meat is the (p+bias) leading block of the weighted moments
        const T factor = (se_type == PDS_HC1) ? nf / (T)(n_rows - pp) : (T)1;
        for (int i = 0; i < pp; ++i) {
            double acc = 0.0;
            for (int a = 0; a < pp; ++a) {
                double t = 0.0;
                for (int b = 0; b < pp; ++b) t += (double)meat[a + (size_t)b * q] * (double)inv[b + (size_t)i * pp];
                acc += (double)inv[a + (size_t)i * pp] * t;
            }
            se[i] = (T)std::sqrt((double)((T)acc * factor));
        }
    }
    const double t_alpha = student_t_ppf(0.975, (double)dof);
    for (int i = 0; i < pp; ++i) {
        out->beta[i] = beta[i];
        out->std_err[i] = se[i];
        const T tv = beta[i] / se[i];
        out->t[i] = tv;
        bool err = false;
        const double sf = student_t_sf(std::fabs((double)tv), (double)dof, &err);
        out->p[i] = err ? (T)NAN : (T)(2.0 * sf);
        out->ci_lower[i] = (T)((double)beta[i] - t_alpha * (double)se[i]);
        out->ci_upper[i] = (T)((double)beta[i] + t_alpha * (double)se[i]);
    }
}

template <typename T, typename R>
static int report_impl(pds_ctx* ctx, const T* const* cols, const T* weights, int n_feat, int64_t n_rows,
                       pds_space space, int add_bias, int se_type, T y_var, R* out,
                       // nullable form (pl_lin_reg_report with a null policy): Arrow validity per column [y, x1..xp]
                       bool nullable = false, const uint8_t* const* validity = nullptr, const int64_t* bit_offsets = nullptr,
                       int policy = PDS_NULL_RAISE, T fill_value = T(0), int64_t* n_used = nullptr) {
    if (!ctx || !cols || !out) return fail(PDS_ERR_INVALID, "null argument");
    if (nullable) {
        if (weights) return fail(PDS_ERR_UNSUPPORTED, "wls_report takes null-free inputs (the reference does not compact its weights)");
        if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
        if (n_rows == 0) return fail(PDS_ERR_EMPTY, "Empty data");
        if (policy < PDS_NULL_RAISE || policy > PDS_NULL_IGNORE) return fail(PDS_ERR_INVALID, "Invalid NullPolicy.");
    } else if (int rc = check_shape(n_feat, n_rows, add_bias)) {
        return rc;
    }
    if (weights && se_type != PDS_SE) se_type = PDS_SE;  // pl_wls_report only knows "std_err"
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int p = n_feat, bias = add_bias ? 1 : 0, pp = p + bias, q = p + 2;
    size_t need = 131072 + sizeof(T) * (size_t)(2 * q * q + pp * pp + pp) + sizeof(T*) * (size_t)(p + 64);
    if (p > kMaxFeatSmall) need += (se_type != PDS_SE ? 2 : 1) * moments_wide_workspace(ctx->num_cus, p, n_rows, true);
    if (se_type != PDS_SE) need += (size_t)n_rows * sizeof(T) + 512;
    if (nullable) {
        need += (1 << 20) + null_policy_workspace(p + 1, n_rows, sizeof(T));
        if (space == PDS_HOST) need += (size_t)(p + 1) * ((size_t)n_rows / 8 + 4096);
    }
    if (int rc = ws_reserve(ctx, need)) return rc;
    DeviceCols<T> dc;
    if (int rc = make_device_cols<T>(ctx, cols, weights, p, n_rows, space, dc)) return rc;
    if (nullable) {
        const int nc = p + 1;
        std::vector<const T*> ref_order(nc);
        ref_order[0] = dc.h_ptrs[p];
        for (int c = 0; c < p; ++c) ref_order[c + 1] = dc.h_ptrs[c];
        std::vector<const uint8_t*> bms(nc, nullptr);
        std::vector<int64_t> boff(nc, 0);
        for (int c = 0; c < nc; ++c) {
            boff[c] = bit_offsets ? bit_offsets[c] : 0;
            const uint8_t* b = validity ? validity[c] : nullptr;
            if (b && space == PDS_HOST) {
                const size_t bytes = (size_t)((boff[c] + n_rows + 7) / 8);
                uint8_t* d = reinterpret_cast<uint8_t*>(ws_take(ctx, bytes));
                PDS_HIP_CHECK(hipMemcpyAsync(d, b, bytes, hipMemcpyHostToDevice, ctx->stream));
                b = d;
            }
            bms[c] = b;
        }
        NullPrepared<T> prep;
        if (int rc = apply_null_policy<T>(ctx, ref_order, bms, boff, n_rows, policy, fill_value, prep)) return rc;
        if (n_used) *n_used = prep.n_kept;
        if (prep.n_kept == 0) return fail(PDS_ERR_EMPTY, "Empty data");
        if (prep.n_kept < pp) return fail(PDS_ERR_TOO_FEW_ROWS, "#Data < #features. No conclusive result.");
        n_rows = prep.n_kept;  // everything below works on the rows that survive the policy
        DeviceCols<T> dk;
        dk.nc = nc;
        dk.h_ptrs.resize(nc);
        for (int c = 0; c < p; ++c) dk.h_ptrs[c] = prep.cols[c + 1];
        dk.h_ptrs[p] = prep.cols[0];
        dk.h_ptrs.resize(std::max(nc, 18), dk.h_ptrs[0]);
        dk.d_ptrs = reinterpret_cast<const T**>(ws_take(ctx, sizeof(T*) * dk.h_ptrs.size()));
        PDS_HIP_CHECK(hipMemcpyAsync(dk.d_ptrs, dk.h_ptrs.data(), sizeof(T*) * dk.h_ptrs.size(), hipMemcpyHostToDevice, ctx->stream));
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // dk.h_ptrs is copied from before dc takes it over
        dc = dk;
    }
    T* d_mom = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * q * q));
    T* d_mom2 = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * q * q));
    T* d_beta = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * (pp + 2)));
    T* d_inv = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * pp * pp));
    uint8_t* d_flag = reinterpret_cast<uint8_t*>(ws_take(ctx, 16));
    double* d_sums = reinterpret_cast<double*>(ws_take(ctx, 64));
    const bool weighted = weights != nullptr;
    if (int rc = launch_moments<T>(ctx, dc, p, n_rows, weighted, d_mom)) return rc;
    // xtx.col_piv_qr() -> inverse() and the solve (:855-858, 1028-1030)
    SolveParams sp{p, bias, PDS_SOLVER_QR, 0.0, 0.0, 0};
    if (int rc = launch_solve<T>(ctx, d_mom, 1, sp, d_beta, d_flag, d_inv, nullptr)) return rc;
    if (int rc = report_second_pass<T>(ctx, dc, p, n_rows, bias, weighted, se_type, d_beta, d_inv, d_sums, d_mom2)) return rc;
    const bool hc = se_type != PDS_SE;
    std::vector<T> beta(pp), inv((size_t)pp * pp), meat((size_t)q * q);
    double sums[2] = {0, 0};
    PDS_HIP_CHECK(hipMemcpyAsync(beta.data(), d_beta, sizeof(T) * pp, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipMemcpyAsync(inv.data(), d_inv, sizeof(T) * pp * pp, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipMemcpyAsync(sums, d_sums, sizeof(sums), hipMemcpyDeviceToHost, ctx->stream));
    if (hc) PDS_HIP_CHECK(hipMemcpyAsync(meat.data(), d_mom2, sizeof(T) * q * q, hipMemcpyDeviceToHost, ctx->stream));
    // y_var = NaN: take the target's sample variance (ddof = 1, what Polars evaluates as `target.var()` and hands over as input
    // 0, expr_linear.py:614-617) from the Gram pass this call has just made -- sum y and sum y^2 are entries of the moment matrix
    T mom_y[2] = {T(0), T(0)};
    const bool derive_var = !(y_var == y_var) && !weighted;
    if (derive_var) {
        PDS_HIP_CHECK(hipMemcpyAsync(&mom_y[0], d_mom + p + (size_t)(p + 1) * q, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipMemcpyAsync(&mom_y[1], d_mom + (p + 1) + (size_t)(p + 1) * q, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    }
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (derive_var) {
        const double nn = (double)n_rows, sy = (double)mom_y[0], syy = (double)mom_y[1];
        y_var = (T)((syy - sy * sy / nn) / (nn - 1.0));
    }
    report_epilogue<T, R>(n_rows, p, bias, se_type, weighted, y_var, beta.data(), inv.data(), meat.data(), sums, out);
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// Row-sharded lin_reg_report: the stages of report_impl as separate entry points, the two exchange
// steps between them (all-reduce of the moment block, all-reduce of [sum e^2 | sum w e^2 | meat]) left to the caller.
// ---------------------------------------------------------------------------------------------
template <typename T>
static int report_fit_impl(pds_ctx* ctx, const T* moments, int n_feat, int add_bias, T* beta, T* inv) {
    if (!ctx || !moments || !beta || !inv) return fail(PDS_ERR_INVALID, "null argument");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int p = n_feat, bias = add_bias ? 1 : 0, pp = p + bias, q = p + 2;
    if (int rc = ws_reserve(ctx, 131072 + sizeof(T) * (size_t)(q * q + pp * pp + pp + 8))) return rc;
    T* d_mom = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * q * q));
    T* d_beta = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * (pp + 2)));
    T* d_inv = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * pp * pp));
    uint8_t* d_flag = reinterpret_cast<uint8_t*>(ws_take(ctx, 16));
    PDS_HIP_CHECK(hipMemcpyAsync(d_mom, moments, sizeof(T) * q * q, hipMemcpyHostToDevice, ctx->stream));
    SolveParams sp{p, bias, PDS_SOLVER_QR, 0.0, 0.0, 0};
    if (int rc = launch_solve<T>(ctx, d_mom, 1, sp, d_beta, d_flag, d_inv, nullptr)) return rc;
    PDS_HIP_CHECK(hipMemcpyAsync(beta, d_beta, sizeof(T) * pp, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipMemcpyAsync(inv, d_inv, sizeof(T) * pp * pp, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

template <typename T>
static int report_partials_impl(pds_ctx* ctx, const T* const* cols, const T* weights, int n_feat, int64_t n_rows, pds_space space,
                                int add_bias, int se_type, const T* beta, const T* inv, double* partials) {
    if (!ctx || !cols || !beta || !inv || !partials) return fail(PDS_ERR_INVALID, "null argument");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    if (weights && se_type != PDS_SE) se_type = PDS_SE;
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int p = n_feat, bias = add_bias ? 1 : 0, pp = p + bias, q = p + 2;
    size_t need = 131072 + sizeof(T) * (size_t)(q * q + pp * pp + pp + 8) + sizeof(T*) * (size_t)(p + 64);
    if (p > kMaxFeatSmall) need += moments_wide_workspace(ctx->num_cus, p, n_rows, true);
    if (se_type != PDS_SE) need += (size_t)n_rows * sizeof(T) + 512;
    if (int rc = ws_reserve(ctx, need)) return rc;
    DeviceCols<T> dc;
    if (int rc = make_device_cols<T>(ctx, cols, weights, p, n_rows, space, dc)) return rc;
    T* d_mom2 = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * q * q));
    T* d_beta = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * (pp + 2)));
    T* d_inv = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * pp * pp));
    double* d_sums = reinterpret_cast<double*>(ws_take(ctx, 64));
    PDS_HIP_CHECK(hipMemcpyAsync(d_beta, beta, sizeof(T) * pp, hipMemcpyHostToDevice, ctx->stream));
    PDS_HIP_CHECK(hipMemcpyAsync(d_inv, inv, sizeof(T) * pp * pp, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = report_second_pass<T>(ctx, dc, p, n_rows, bias, weights != nullptr, se_type, d_beta, d_inv, d_sums, d_mom2)) return rc;
    std::vector<T> meat((size_t)q * q, T(0));
    PDS_HIP_CHECK(hipMemcpyAsync(partials, d_sums, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (se_type != PDS_SE) PDS_HIP_CHECK(hipMemcpyAsync(meat.data(), d_mom2, sizeof(T) * q * q, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < q * q; ++i) partials[2 + i] = (double)meat[i];
    return PDS_OK;
}

template <typename T, typename R>
static int report_finish_impl(int n_feat, int add_bias, int se_type, int weighted, int64_t n_rows_total, T y_var, const T* beta,
                              const T* inv, const double* partials, R* out) {
    if (!beta || !inv || !partials || !out) return fail(PDS_ERR_INVALID, "null argument");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (weighted && se_type != PDS_SE) se_type = PDS_SE;
    const int q = n_feat + 2;
    std::vector<T> meat((size_t)q * q);
    for (int i = 0; i < q * q; ++i) meat[i] = (T)partials[2 + i];
    report_epilogue<T, R>(n_rows_total, n_feat, add_bias ? 1 : 0, se_type, weighted != 0, y_var, beta, inv, meat.data(), partials, out);
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// grouped
// ---------------------------------------------------------------------------------------------
// Groups with more than 64 features (coverage path): every group's Gram matrix comes from the tiled matrix-core SYRK of the
// single-regression path (moments_wide.hip) on that group's row range, the records of a chunk of groups are then solved
// together (solve_big.hip: Cholesky on an L2-resident workspace, one workgroup per system; CD / NNLS: one wavefront each).
__global__ void mark_small_groups_kernel(const int64_t* __restrict__ off, int64_t n_groups, int pp, uint8_t* __restrict__ flags) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_groups && off[g + 1] - off[g] < pp) flags[g] = 1;
}
template <typename T>
__global__ void nan_flagged_kernel(const uint8_t* __restrict__ flags, int64_t n_groups, int pp, T* __restrict__ coeffs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_groups * pp && flags[i / pp]) coeffs[i] = (T)__builtin_nan("");
}

template <typename T>
static int grouped_big(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, const int64_t* d_off, int64_t n_groups, int64_t chunk,
                       const Method& method, const pds_lr_params* prm, const SolveParams& sp, T* d_mom, T* d_coeffs,
                       uint8_t* d_null) {
    const int bias = prm->add_bias ? 1 : 0, pp = n_feat + bias, q = n_feat + 2, nc = n_feat + 1;
    std::vector<int64_t> off((size_t)n_groups + 1);
    PDS_HIP_CHECK(hipMemcpyAsync(off.data(), d_off, off.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // one pointer table per group: the column bases moved to the group's first row
    std::vector<const T*> tables((size_t)n_groups * nc);
    for (int64_t g = 0; g < n_groups; ++g)
        for (int c = 0; c < nc; ++c) tables[(size_t)g * nc + c] = dc.h_ptrs[c] + off[g];
    const T** d_tables = reinterpret_cast<const T**>(ws_take(ctx, tables.size() * sizeof(T*)));
    PDS_HIP_CHECK(hipMemcpyAsync(d_tables, tables.data(), tables.size() * sizeof(T*), hipMemcpyHostToDevice, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const bool f32 = sizeof(T) == 4;
    for (int64_t g0 = 0; g0 < n_groups; g0 += chunk) {
        const int64_t gc = std::min(chunk, n_groups - g0);
        for (int64_t g = g0; g < g0 + gc; ++g) {
            T* rec = d_mom + (size_t)(g - g0) * q * q;
            const int64_t ng = off[g + 1] - off[g];
            if (ng <= 0) {
                PDS_HIP_CHECK(hipMemsetAsync(rec, 0, sizeof(T) * (size_t)q * q, ctx->stream));
                continue;
            }
            DeviceCols<T> dg;
            dg.nc = nc;
            dg.d_ptrs = d_tables + (size_t)g * nc;
            dg.h_ptrs.assign(tables.begin() + (size_t)g * nc, tables.begin() + (size_t)(g + 1) * nc);
            const size_t mark = ctx->ws_used;  // the SYRK partials are call-local: stream order makes the reuse safe
            const int rc = launch_moments_wide<T>(ctx, dg, n_feat, ng, false, rec);
            ctx->ws_used = mark;
            if (rc) return rc;
        }
        T* co = d_coeffs + g0 * pp;
        uint8_t* fl = d_null + g0;
        if (method.kind == Method::OLS) {
            if (int rc = launch_solve<T>(ctx, d_mom, gc, sp, co, fl, nullptr, nullptr)) return rc;
            // per-group pl_lr rejects "#Data < #features": null
            hipLaunchKernelGGL(mark_small_groups_kernel, dim3((unsigned)((gc + 255) / 256)), dim3(256), 0, ctx->stream, d_off + g0, gc,
                               pp, fl);
            hipLaunchKernelGGL((nan_flagged_kernel<T>), dim3((unsigned)((gc * pp + 255) / 256)), dim3(256), 0, ctx->stream, fl, gc, pp,
                               co);
            PDS_HIP_CHECK(hipGetLastError());
        } else if (method.kind == Method::NNLS) {
            if (int rc = launch_nnls<T>(ctx, d_mom, n_feat, bias, prm->tol, f32 ? 200 : prm->max_iter, co, gc, fl, d_off + g0)) return rc;
        } else if (int rc = launch_cd<T>(ctx, d_mom, n_feat, bias, method.l1, method.l2, prm->tol, f32 ? 2000 : prm->max_iter,
                                         method.positive, co, nullptr, gc, fl, d_off + g0))
            return rc;
    }
    return PDS_OK;
}

template <typename T>
static int grouped_impl(pds_ctx* ctx, const T* const* cols, int n_feat, int64_t n_rows, const int64_t* offsets,
                        int64_t n_groups, pds_space space, const pds_lr_params* prm, T* coeffs, uint8_t* is_null,
                        // nullable form: Arrow validity per column [y, x1..xp]; every group is fitted on the rows of it that
                        // survive the policy, like Polars calling pl_lr(null_policy=...) per group
                        bool nullable = false, const uint8_t* const* validity = nullptr, const int64_t* bit_offsets = nullptr,
                        int policy = PDS_NULL_RAISE, T fill_value = T(0)) {
    if (!ctx || !cols || !offsets || !prm || !coeffs) return fail(PDS_ERR_INVALID, "null argument");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_groups <= 0 || n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    const Method method = pick_method(prm);  // per group what pl_lr does per call: linear_regression.rs:447-497
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int bias = prm->add_bias ? 1 : 0, pp = n_feat + bias, q = n_feat + 2;
    // chunk the groups so one chunk's moment records (q*q values per group) stay inside the 256 MiB
    // Infinity Cache between the Gram kernel that writes them and the solve kernel that reads them
    const bool big = n_feat > kMaxFeatWide;  // > 64 features: one tiled-SYRK Gram build per group + solve_big
    int64_t chunk = std::max<int64_t>(big ? 64 : 4096, (int64_t)(128ll << 20) / (int64_t)(sizeof(T) * q * q));
    chunk = std::min(chunk, n_groups);
    size_t need = 131072 + sizeof(T) * (size_t)chunk * q * q;
    need += (size_t)n_groups * 4 + (size_t)chunk * (pp * sizeof(T) + 1) + 4096;  // the fused path's pivoted-QR pass: list, results
    if (big) need += (size_t)n_groups * (n_feat + 1) * sizeof(T*) + moments_wide_workspace(ctx->num_cus, n_feat, n_rows) + 8192;
    if (space == PDS_HOST) need += (size_t)(n_groups + 1) * 8 + (size_t)n_groups * (pp * sizeof(T) + 1) + 4096;
    if (nullable) {
        if (policy < PDS_NULL_RAISE || policy > PDS_NULL_IGNORE) return fail(PDS_ERR_INVALID, "Invalid NullPolicy.");
        need += (1 << 20) + null_policy_workspace(n_feat + 1, n_rows, sizeof(T)) + (size_t)(n_groups + 1) * 8 + 4096;
        if (space == PDS_HOST) need += (size_t)(n_feat + 1) * ((size_t)n_rows / 8 + 4096);
    }
    if (int rc = ws_reserve(ctx, need)) return rc;
    DeviceCols<T> dc;
    if (int rc = make_device_cols<T>(ctx, cols, (const T*)nullptr, n_feat, n_rows, space, dc)) return rc;
    const int64_t* d_off = offsets;
    T* d_coeffs = coeffs;
    uint8_t* d_null = is_null;
    // small host batches (the plugin layer's coalesced per-group calls): offsets go up through pinned memory, coefficients
    // and null flags come back in ONE copy -- every pageable hipMemcpyAsync is 5-10 us of a ~60 us call
    const size_t co_bytes = ((size_t)n_groups * pp * sizeof(T) + 255) & ~(size_t)255;
    const bool small_out = space == PDS_HOST && co_bytes + (size_t)n_groups + (size_t)(n_groups + 1) * 8 <= ((size_t)48 << 10);
    if (space == PDS_HOST) {
        int64_t* t = reinterpret_cast<int64_t*>(ws_take(ctx, (size_t)(n_groups + 1) * 8));
        if (small_out) {
            if (int rc = ensure_pinned(ctx, (size_t)128 << 10)) return rc;
            char* pin_off = static_cast<char*>(ctx->pinned) + ((size_t)64 << 10);
            std::memcpy(pin_off, offsets, (size_t)(n_groups + 1) * 8);
            PDS_HIP_CHECK(hipMemcpyAsync(t, pin_off, (size_t)(n_groups + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        } else {
            PDS_HIP_CHECK(hipMemcpyAsync(t, offsets, (size_t)(n_groups + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        }
        d_off = t;
        char* blk = reinterpret_cast<char*>(ws_take(ctx, co_bytes + (size_t)n_groups));
        d_coeffs = reinterpret_cast<T*>(blk);
        d_null = reinterpret_cast<uint8_t*>(blk + co_bytes);
    } else if (!d_null) {
        d_null = reinterpret_cast<uint8_t*>(ws_take(ctx, (size_t)n_groups));
    }
    if (nullable) {
        const int nc = n_feat + 1;
        std::vector<const T*> ref_order(nc);
        ref_order[0] = dc.h_ptrs[n_feat];
        for (int c = 0; c < n_feat; ++c) ref_order[c + 1] = dc.h_ptrs[c];
        std::vector<const uint8_t*> bms(nc, nullptr);
        std::vector<int64_t> boff(nc, 0);
        for (int c = 0; c < nc; ++c) {
            boff[c] = bit_offsets ? bit_offsets[c] : 0;
            const uint8_t* b = validity ? validity[c] : nullptr;
            if (b && space == PDS_HOST) {
                const size_t bytes = (size_t)((boff[c] + n_rows + 7) / 8);
                uint8_t* d = reinterpret_cast<uint8_t*>(ws_take(ctx, bytes));
                PDS_HIP_CHECK(hipMemcpyAsync(d, b, bytes, hipMemcpyHostToDevice, ctx->stream));
                b = d;
            }
            bms[c] = b;
        }
        NullPrepared<T> prep;
        if (int rc = apply_null_policy<T>(ctx, ref_order, bms, boff, n_rows, policy, fill_value, prep)) return rc;
        if (prep.dropped) {
            int64_t* off2 = reinterpret_cast<int64_t*>(ws_take(ctx, (size_t)(n_groups + 1) * 8));
            if (int rc = remap_group_offsets(ctx, d_off, n_groups, prep.d_rank, n_rows, prep.n_kept, off2)) return rc;
            d_off = off2;
        }
        DeviceCols<T> dk;
        dk.nc = nc;
        dk.h_ptrs.resize(nc);
        for (int c = 0; c < n_feat; ++c) dk.h_ptrs[c] = prep.cols[c + 1];
        dk.h_ptrs[n_feat] = prep.cols[0];
        dk.h_ptrs.resize(std::max(nc, 18), dk.h_ptrs[0]);
        dk.d_ptrs = reinterpret_cast<const T**>(ws_take(ctx, sizeof(T*) * dk.h_ptrs.size()));
        PDS_HIP_CHECK(hipMemcpyAsync(dk.d_ptrs, dk.h_ptrs.data(), sizeof(T*) * dk.h_ptrs.size(), hipMemcpyHostToDevice, ctx->stream));
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        dc = dk;
        n_rows = prep.n_kept;
        if (n_rows == 0) return fail(PDS_ERR_EMPTY, "Empty data");
    }
    T* d_mom = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * (size_t)chunk * q * q));
    SolveParams sp{n_feat, bias, prm->solver == PDS_SOLVER_SVD ? PDS_SOLVER_QR : prm->solver, prm->l2_reg,
                   prm->singular_x_tol, 0};
    {
        const char* piv0 = std::getenv("PDS_GROUPED_PIVOTED");
        if (piv0 && piv0[0] == '1' && sp.solver == PDS_SOLVER_CHOLESKEY) sp.solver = PDS_SOLVER_QR;
    }
    // Default (rank gate on): ONE streaming kernel, Gram + in-register Cholesky, no moment records in HBM.
    // Gate off (singular_x_tol = 0) needs the pivoted QR to reproduce the reference's answers on rank-deficient
    // groups; that solver is register hungry and runs faster as its own kernel behind the grouped Gram build.
    // PDS_GROUPED_UNFUSED=1 / PDS_GROUPED_PIVOTED=1 force the two-kernel pipeline / the pivoted QR (development).
    const char* unfused_env = std::getenv("PDS_GROUPED_UNFUSED");
    const char* piv_env = std::getenv("PDS_GROUPED_PIVOTED");
    const bool want_piv = !(sp.gate_tol > 0.0) || (piv_env && piv_env[0] == '1');
    if (big) {
        if (int rc = grouped_big<T>(ctx, dc, n_feat, d_off, n_groups, chunk, method, prm, sp, d_mom, d_coeffs, d_null)) return rc;
    } else if (method.kind != Method::OLS) {
        // lasso / elastic net / positive fits per group: grouped Gram build, then one wavefront per group runs the
        // reference's coordinate descent (faer_coordinate_descent / faer_nn_lr) on that group's moment record
        const bool f32 = sizeof(T) == 4;
        for (int64_t g0 = 0; g0 < n_groups; g0 += chunk) {
            const int64_t gc = std::min(chunk, n_groups - g0);
            if (int rc = launch_grouped_moments<T>(ctx, dc, n_feat, d_off + g0, gc, d_mom)) return rc;
            if (method.kind == Method::NNLS) {
                if (int rc = launch_nnls<T>(ctx, d_mom, n_feat, bias, prm->tol, f32 ? 200 : prm->max_iter, d_coeffs + g0 * pp, gc,
                                            d_null + g0, d_off + g0))
                    return rc;
            } else if (int rc = launch_cd<T>(ctx, d_mom, n_feat, bias, method.l1, method.l2, prm->tol,
                                             f32 ? 2000 : prm->max_iter, method.positive, d_coeffs + g0 * pp, nullptr, gc,
                                             d_null + g0, d_off + g0))
                return rc;
        }
    } else if (n_feat <= 16 && !want_piv && !(unfused_env && unfused_env[0] == '1') && n_groups < (1ll << 31)) {
        if (int rc = launch_grouped_fused<T>(ctx, dc, n_feat, n_rows, d_off, n_groups, sp, d_coeffs, d_null, d_mom, chunk)) return rc;
    } else {
        for (int64_t g0 = 0; g0 < n_groups; g0 += chunk) {
            const int64_t gc = std::min(chunk, n_groups - g0);
            if (int rc = launch_grouped_moments<T>(ctx, dc, n_feat, d_off + g0, gc, d_mom)) return rc;
            if (int rc = launch_solve<T>(ctx, d_mom, gc, sp, d_coeffs + g0 * pp, d_null + g0, nullptr, d_off + g0)) return rc;
        }
    }
    if (small_out) {
        char* pin = static_cast<char*>(ctx->pinned);
        PDS_HIP_CHECK(hipMemcpyAsync(pin, d_coeffs, co_bytes + (size_t)n_groups, hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        std::memcpy(coeffs, pin, (size_t)n_groups * pp * sizeof(T));
        if (is_null) std::memcpy(is_null, pin + co_bytes, (size_t)n_groups);
        return PDS_OK;
    }
    if (space == PDS_HOST) {
        PDS_HIP_CHECK(hipMemcpyAsync(coeffs, d_coeffs, (size_t)n_groups * pp * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        if (is_null) PDS_HIP_CHECK(hipMemcpyAsync(is_null, d_null, (size_t)n_groups, hipMemcpyDeviceToHost, ctx->stream));
    }
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// faer_qr_lr_with_inv (lr_online_solvers.rs:120-143): the initial fit of OnlineLR -- coefficients and (X'X + lambda)^-1
// ---------------------------------------------------------------------------------------------
template <typename T>
static int lr_with_inv_impl(pds_ctx* ctx, const T* const* cols, int n_feat, int64_t n_rows, pds_space space, int add_bias,
                            double lambda, T* coeffs, T* inv) {
    if (!ctx || !cols || !coeffs || !inv) return fail(PDS_ERR_INVALID, "null argument");
    if (int rc = check_shape(n_feat, n_rows, add_bias)) return rc;
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int bias = add_bias ? 1 : 0, pp = n_feat + bias, q = n_feat + 2;
    size_t need = 65536 + sizeof(T) * ((size_t)q * q + (size_t)pp * pp + pp + 8) + sizeof(T*) * (size_t)(n_feat + 32);
    if (n_feat > kMaxFeatSmall) need += moments_wide_workspace(ctx->num_cus, n_feat, n_rows);
    if (int rc = ws_reserve(ctx, need)) return rc;
    DeviceCols<T> dc;
    if (int rc = make_device_cols<T>(ctx, cols, (const T*)nullptr, n_feat, n_rows, space, dc)) return rc;
    T* d_mom = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * q * q));
    T* d_beta = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * (pp + 2)));
    T* d_inv = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * pp * pp));
    uint8_t* d_flag = reinterpret_cast<uint8_t*>(ws_take(ctx, 16));
    if (int rc = launch_moments<T>(ctx, dc, n_feat, n_rows, false, d_mom)) return rc;
    SolveParams sp{n_feat, bias, PDS_SOLVER_QR, lambda > 0.0 ? lambda : 0.0, 0.0, 0};
    if (int rc = launch_solve<T>(ctx, d_mom, 1, sp, d_beta, d_flag, d_inv, nullptr)) return rc;
    PDS_HIP_CHECK(hipMemcpyAsync(coeffs, d_beta, sizeof(T) * pp, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipMemcpyAsync(inv, d_inv, sizeof(T) * pp * pp, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// weighted groups: per group faer_weighted_lr (lr_solvers.rs:386-409) -- X' W X = (sqrt(W) X)' (sqrt(W) X), so the frame
// is scaled once on the device (the bias becomes an explicit sqrt(w) column) and takes the unweighted, ungated grouped path
// ---------------------------------------------------------------------------------------------
template <typename T>
static int grouped_weighted_impl(pds_ctx* ctx, const T* const* cols, const T* weights, int n_feat, int64_t n_rows,
                                 const int64_t* offsets, int64_t n_groups, pds_space space, const pds_lr_params* prm, T* coeffs,
                                 uint8_t* is_null) {
    if (!ctx || !cols || !weights || !offsets || !prm || !coeffs) return fail(PDS_ERR_INVALID, "null argument");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_groups <= 0 || n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int bias = prm->add_bias ? 1 : 0, pf = n_feat + bias, nc_in = n_feat + 1;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t col_bytes = up((size_t)n_rows * sizeof(T));
    size_t need = col_bytes * (pf + 1) + 4096;
    if (space == PDS_HOST) need += col_bytes * (nc_in + 1) + up((size_t)(n_groups + 1) * 8) + up((size_t)n_groups * pf * sizeof(T)) + up((size_t)n_groups);
    if (int rc = ensure_ws(ctx, ctx->keyed, need)) return rc;
    char* w = static_cast<char*>(ctx->keyed.ptr);
    auto take = [&](size_t b) { char* r = w; w += up(b); return r; };
    std::vector<const T*> src(nc_in);
    const T* d_w = weights;
    const int64_t* d_off = offsets;
    T* d_co = coeffs;
    uint8_t* d_nu = is_null;
    if (space == PDS_HOST) {
        for (int c = 0; c < nc_in; ++c) {
            T* d = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
            PDS_HIP_CHECK(hipMemcpyAsync(d, cols[c], (size_t)n_rows * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
            src[c] = d;
        }
        T* dw = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
        PDS_HIP_CHECK(hipMemcpyAsync(dw, weights, (size_t)n_rows * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        d_w = dw;
        int64_t* doff = reinterpret_cast<int64_t*>(take((size_t)(n_groups + 1) * 8));
        PDS_HIP_CHECK(hipMemcpyAsync(doff, offsets, (size_t)(n_groups + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        d_off = doff;
        d_co = reinterpret_cast<T*>(take((size_t)n_groups * pf * sizeof(T)));
        d_nu = reinterpret_cast<uint8_t*>(take((size_t)n_groups));
    } else {
        for (int c = 0; c < nc_in; ++c) src[c] = cols[c];
    }
    // scaled frame in reference order [y, x1..xp, (sqrt w)]
    std::vector<const T*> scaled(pf + 1);
    for (int c = 0; c < nc_in; ++c) {
        T* d = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
        if (int rc = launch_scale_sqrt_w<T>(ctx, src[c], d_w, n_rows, d)) return rc;
        scaled[c] = d;
    }
    if (bias) {
        T* d = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
        if (int rc = launch_scale_sqrt_w<T>(ctx, (const T*)nullptr, d_w, n_rows, d)) return rc;
        scaled[nc_in] = d;
    }
    pds_lr_params p2 = *prm;  // faer_weighted_lr: plain solve with `solver`, no gate, no penalties
    p2.add_bias = 0;
    p2.l1_reg = 0.0;
    p2.l2_reg = 0.0;
    p2.positive = 0;
    p2.singular_x_tol = 0.0;
    if (int rc = grouped_impl<T>(ctx, scaled.data(), pf, n_rows, d_off, n_groups, PDS_DEVICE, &p2, d_co, d_nu)) return rc;
    if (space == PDS_HOST) {
        PDS_HIP_CHECK(hipMemcpyAsync(coeffs, d_co, (size_t)n_groups * pf * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        if (is_null) PDS_HIP_CHECK(hipMemcpyAsync(is_null, d_nu, (size_t)n_groups, hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// grouped by an int64 key column in any row order (keyed.hip brings the frame into key order on the device)
// ---------------------------------------------------------------------------------------------
template <typename T>
static int lr_by_key_impl(pds_ctx* ctx, const T* const* cols, const int64_t* keys, int n_feat, int64_t n_rows, pds_space space,
                          const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, T* coeffs, uint8_t* is_null,
                          int64_t* n_groups) {
    if (!ctx || !cols || !keys || !prm || !out_keys || !coeffs || !n_groups) return fail(PDS_ERR_INVALID, "null argument");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    if (n_rows >= (1ll << 31)) return fail(PDS_ERR_UNSUPPORTED, "keyed grouping: fewer than 2^31 rows per call");
    if (max_groups < 1) return fail(PDS_ERR_INVALID, "max_groups must be positive");
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int nc = n_feat + 1, pp = n_feat + (prm->add_bias ? 1 : 0);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t key_bytes = up((size_t)n_rows * 8), col_bytes = up((size_t)n_rows * sizeof(T)), idx_bytes = up((size_t)n_rows * 4);
    // ---- keys on the device, and are they already in order?
    const int64_t* d_keys = keys;
    if (space == PDS_HOST) {
        if (int rc = ensure_ws(ctx, ctx->stage, key_bytes + 256)) return rc;
        PDS_HIP_CHECK(hipMemcpyAsync(ctx->stage.ptr, keys, (size_t)n_rows * 8, hipMemcpyHostToDevice, ctx->stream));
        d_keys = static_cast<const int64_t*>(ctx->stage.ptr);
    }
    if (int rc = ensure_pinned(ctx, 4096)) return rc;
    if (int rc = ensure_ws(ctx, ctx->solve_ws, 4096)) return rc;  // a flag word that outlives the workspace sizing below
    bool sorted = false;
    if (int rc = keys_nondecreasing(ctx, d_keys, n_rows, static_cast<unsigned*>(ctx->solve_ws.ptr), &sorted)) return rc;
    // ---- workspace: [raw columns (host frames)] [sorted keys, index in/out, gathered columns (unsorted frames)] runs, temp
    const int64_t cap = std::min<int64_t>(max_groups, n_rows);
    const size_t temp_bytes = keyed_temp_bytes(n_rows);
    size_t need = temp_bytes + 3 * up((size_t)(n_rows + 1) * 8) + 4096;  // unique keys, counts, offsets (at most one per row)
    if (space == PDS_HOST) need += col_bytes * nc + up((size_t)cap * pp * sizeof(T)) + up((size_t)cap);
    if (!sorted) need += 2 * key_bytes + 2 * idx_bytes + col_bytes * nc + up((size_t)n_rows * nc * sizeof(T)) + up(2 * (size_t)nc * sizeof(T*)) + 1024;
    if (int rc = ensure_ws(ctx, ctx->keyed, need)) return rc;
    char* w = static_cast<char*>(ctx->keyed.ptr);
    auto take = [&](size_t b) { char* r = w; w += up(b); return r; };
    void* d_temp = take(temp_bytes);
    int64_t* d_unique = reinterpret_cast<int64_t*>(take((size_t)(n_rows + 1) * 8));
    int64_t* d_counts = reinterpret_cast<int64_t*>(take((size_t)(n_rows + 1) * 8));
    int64_t* d_offsets = reinterpret_cast<int64_t*>(take((size_t)(n_rows + 1) * 8));
    int64_t* d_nruns = reinterpret_cast<int64_t*>(take(256));
    std::vector<const T*> src(nc);  // reference order [y, x1..xp], device resident
    for (int c = 0; c < nc; ++c) src[c] = cols[c];
    if (space == PDS_HOST)
        for (int c = 0; c < nc; ++c) {
            T* d = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
            PDS_HIP_CHECK(hipMemcpyAsync(d, cols[c], (size_t)n_rows * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
            src[c] = d;
        }
    const int64_t* d_sorted_keys = d_keys;
    if (!sorted) {
        int64_t* sk = reinterpret_cast<int64_t*>(take((size_t)n_rows * 8));
        uint32_t* idx_in = reinterpret_cast<uint32_t*>(take((size_t)n_rows * 4));
        uint32_t* perm = reinterpret_cast<uint32_t*>(take((size_t)n_rows * 4));
        int64_t* sk2 = reinterpret_cast<int64_t*>(take((size_t)n_rows * 8));
        int64_t* mm = reinterpret_cast<int64_t*>(take(256));
        if (int rc = keyed_sort(ctx, d_keys, n_rows, idx_in, sk, perm, d_temp, temp_bytes, sk2, mm)) return rc;
        d_sorted_keys = sk;
        static const bool by_column = [] { const char* e = std::getenv("PDS_KEYED_GATHER_BY_COLUMN"); return e && e[0] == '1'; }();
        if (by_column) {  // (A/B: one random 8-byte read per element)
            for (int c = 0; c < nc; ++c) {
                T* d = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
                if (int rc = launch_gather_rows<T>(ctx, src[c], perm, n_rows, d)) return rc;
                src[c] = d;
            }
        } else {
            // transpose to row-major records, then one random access per ROW (keyed.hip)
            std::vector<const T*> tbl(2 * (size_t)nc);
            for (int c = 0; c < nc; ++c) tbl[c] = src[c];
            for (int c = 0; c < nc; ++c) {
                T* d = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
                tbl[nc + c] = d;
                src[c] = d;
            }
            T* records = reinterpret_cast<T*>(take((size_t)n_rows * nc * sizeof(T)));
            const T** d_tbl = reinterpret_cast<const T**>(take(2 * (size_t)nc * sizeof(T*)));
            PDS_HIP_CHECK(hipMemcpyAsync(d_tbl, tbl.data(), 2 * (size_t)nc * sizeof(T*), hipMemcpyHostToDevice, ctx->stream));
            if (int rc = launch_gather_frame<T>(ctx, d_tbl, perm, nc, n_rows, records, (T* const*)(d_tbl + nc)))
                return rc;
            PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (tbl: source of the table copy)
        }
    }
    int64_t ng = 0;
    if (int rc = keyed_runs(ctx, d_sorted_keys, n_rows, d_unique, d_counts, d_offsets, d_nruns, d_temp, temp_bytes, &ng)) return rc;
    *n_groups = ng;
    if (ng > max_groups) return fail(PDS_ERR_INVALID, "more distinct keys than max_groups");
    T* d_co = coeffs;
    uint8_t* d_nu = is_null;
    if (space == PDS_HOST) {
        d_co = reinterpret_cast<T*>(take((size_t)cap * pp * sizeof(T)));
        d_nu = reinterpret_cast<uint8_t*>(take((size_t)cap));
    }
    if (int rc = grouped_impl<T>(ctx, src.data(), n_feat, n_rows, d_offsets, ng, PDS_DEVICE, prm, d_co, d_nu)) return rc;
    if (space == PDS_HOST) {
        PDS_HIP_CHECK(hipMemcpyAsync(coeffs, d_co, (size_t)ng * pp * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        if (is_null) PDS_HIP_CHECK(hipMemcpyAsync(is_null, d_nu, (size_t)ng, hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipMemcpyAsync(out_keys, d_unique, (size_t)ng * 8, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        PDS_HIP_CHECK(hipMemcpyAsync(out_keys, d_unique, (size_t)ng * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

template <typename T>
static int rolling_impl(pds_ctx* ctx, const T* const* cols, int n_feat, int64_t n_rows, pds_space space, int add_bias,
                        int64_t window, int64_t min_size, double lambda, bool expanding, T* coeffs, T* pred,
                        uint8_t* valid, const T* seed_moments = nullptr) {
    if (!ctx || !cols || !coeffs || !pred || !valid) return fail(PDS_ERR_INVALID, "null argument");
    if (int rc = check_shape(n_feat, n_rows, add_bias)) return rc;
    const int pp = n_feat + (add_bias ? 1 : 0);
    std::vector<double> seed;
    if (seed_moments) {  // rows in front of this frame: they count towards start_with
        const int q = n_feat + 2;
        seed.assign(seed_moments, seed_moments + (size_t)q * q);
        for (double v : seed)
            if (!std::isfinite(v)) return fail(PDS_ERR_INVALID, "seed moments must be finite");
        const double seen = seed[n_feat + (size_t)n_feat * q];
        if (window < 1 || seen < 0.0) return fail(PDS_ERR_INVALID, "start_with must be >= 1 and the seed row count >= 0");
        const double left = (double)window - seen;
        window = left <= 1.0 ? 1 : (left > (double)n_rows ? n_rows + 1 : (int64_t)left);
    } else if (window < 1 || window > n_rows) {
        return fail(PDS_ERR_INVALID, "window / start_with must be in [1, n_rows]");
    }
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    size_t need = 131072 + ((size_t)(n_rows / 4096) + 2) * 96 * sizeof(double)  // + per-tile totals (expanding)
                  + ((size_t)(n_rows / 4096 / 32) + 2) * 128 * sizeof(double);   // + their chunk sums (tile prefix)
    if (space == PDS_HOST) need += (size_t)n_rows * ((pp + 1) * sizeof(T) + 1) + 4096;
    if (pp > 12) need += rolling_wide_workspace(n_feat, n_rows, sizeof(T));
    if (int rc = ws_reserve(ctx, need)) return rc;
    DeviceCols<T> dc;
    if (int rc = make_device_cols<T>(ctx, cols, (const T*)nullptr, n_feat, n_rows, space, dc)) return rc;
    T* d_co = coeffs;
    T* d_pr = pred;
    uint8_t* d_va = valid;
    if (space == PDS_HOST) {
        d_co = reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * pp * sizeof(T)));
        d_pr = reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T)));
        d_va = reinterpret_cast<uint8_t*>(ws_take(ctx, (size_t)n_rows));
    }
    if (int rc = launch_rolling<T>(ctx, dc, n_feat, n_rows, add_bias, window, min_size, lambda, expanding,
                                   seed.empty() ? nullptr : seed.data(), d_co, d_pr, d_va))
        return rc;
    if (space == PDS_HOST) {
        PDS_HIP_CHECK(hipMemcpyAsync(coeffs, d_co, (size_t)n_rows * pp * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipMemcpyAsync(pred, d_pr, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipMemcpyAsync(valid, d_va, (size_t)n_rows, hipMemcpyDeviceToHost, ctx->stream));
    }
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// Fits straight from a row-major matrix (the pyclass route, src/pymodels/py_lr.rs over numpy_faer.rs:10-66).
// mode 0: LR::fit (faer_solve_lr through the pl_lr dispatch of `prm`), 1: ElasticNet::fit (always coordinate descent),
// 2: OnlineLR::fit (faer_qr_lr_with_inv: coefficients + (X'X + lambda)^-1, lambda = prm->l2_reg).
// Up to 16 features the Gram matrix comes from moments_rowmajor_kernel -- the matrix core reads the rows as they lie, ONE
// pass, nothing transposed; a host matrix crosses PCIe as contiguous row chunks, each chunk one f64 moment record.  Wider
// matrices are transposed once on the device (layout.hip) and take the column path.
// ---------------------------------------------------------------------------------------------
template <typename T>
static int lr_rowmajor_impl(pds_ctx* ctx, const T* X, int64_t ld, const T* y, int64_t n_rows, int n_feat, pds_space space,
                            const pds_lr_params* prm, int mode, T* coeffs, int* is_null, T* inv) {
    if (!ctx || !X || !y || !prm || !coeffs) return fail(PDS_ERR_INVALID, "null argument");
    if (mode < 0 || mode > 2 || (mode == 2 && !inv)) return fail(PDS_ERR_INVALID, "mode: 0 LR, 1 ElasticNet, 2 with inverse (inv required)");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (ld < n_feat) return fail(PDS_ERR_INVALID, "row stride < columns");
    if (n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    if (mode != 1)
        if (int rc = check_shape(n_feat, n_rows, prm->add_bias)) return rc;
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int p = n_feat, bias = prm->add_bias ? 1 : 0, pp = p + bias, q = p + 2;
    if (p > kMaxFeatSmall) {
        // wide: one transposition into column buffers (kept in ctx->keyed), then the column entry points
        const size_t col_bytes = ((size_t)n_rows * sizeof(T) + 255) & ~(size_t)255;
        if (int rc = ensure_ws(ctx, ctx->keyed, col_bytes * (p + 1) + 256)) return rc;
        T* base = reinterpret_cast<T*>(ctx->keyed.ptr);
        const int64_t stride = (int64_t)(col_bytes / sizeof(T));
        if (int rc = rows_to_cols_impl<T>(ctx, X, ld, n_rows, p, space, base + stride, stride)) return rc;  // columns 1..p
        PDS_HIP_CHECK(hipMemcpyAsync(base, y, (size_t)n_rows * sizeof(T), space == PDS_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                                     ctx->stream));
        std::vector<const T*> cols(p + 1);
        for (int c = 0; c <= p; ++c) cols[c] = base + (int64_t)c * stride;  // [y, x1..xp]
        if (mode == 2) return lr_with_inv_impl<T>(ctx, cols.data(), p, n_rows, PDS_DEVICE, bias, prm->l2_reg, coeffs, inv);
        return lr_impl<T>(ctx, cols.data(), (const T*)nullptr, p, n_rows, PDS_DEVICE, prm, coeffs, is_null, (T*)nullptr, (T*)nullptr, mode == 1);
    }
    int64_t rows_per = n_rows;
    int nchunks = 1;
    if (space == PDS_HOST) {
        rows_per = (int64_t)(host_chunk_bytes() / ((size_t)(ld + 1) * sizeof(T)));
        rows_per = std::min<int64_t>(std::max<int64_t>(rows_per & ~(int64_t)31, 32), n_rows);
        nchunks = (int)((n_rows + rows_per - 1) / rows_per);
    }
    if (int rc = ws_reserve(ctx, 131072 + sizeof(T) * ((size_t)q * q + (size_t)pp * pp + pp + 16) + (size_t)nchunks * q * q * sizeof(double))) return rc;
    T* d_mom = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * q * q));
    if (space == PDS_DEVICE) {
        if (int rc = launch_moments_rowmajor<T>(ctx, X, ld, y, p, n_rows, d_mom)) return rc;
    } else {
        if (int rc = ensure_ws(ctx, ctx->stage, ((size_t)rows_per * (ld + 1)) * sizeof(T) + 512)) return rc;
        T* d_X = reinterpret_cast<T*>(ctx->stage.ptr);
        T* d_y = d_X + (size_t)rows_per * ld;
        double* d_slots = reinterpret_cast<double*>(ws_take(ctx, (size_t)nchunks * q * q * sizeof(double)));
        if (!d_slots) return fail(PDS_ERR_HIP, "workspace allocation failed");
        for (int k = 0; k < nchunks; ++k) {
            const int64_t r0 = (int64_t)k * rows_per, rows = std::min(rows_per, n_rows - r0);
            PDS_HIP_CHECK(hipMemcpyAsync(d_X, X + r0 * ld, ((size_t)(rows - 1) * ld + p) * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
            PDS_HIP_CHECK(hipMemcpyAsync(d_y, y + r0, (size_t)rows * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
            if (int rc = launch_moments_rowmajor<T>(ctx, d_X, ld, d_y, p, rows, (T*)nullptr, d_slots + (size_t)k * q * q)) return rc;
        }
        if (int rc = launch_sum_moment_slots<T>(ctx, d_slots, nchunks, q * q, d_mom)) return rc;
    }
    if (mode == 2) {
        T* d_beta = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * (pp + 2)));
        T* d_inv = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * pp * pp));
        uint8_t* d_flag = reinterpret_cast<uint8_t*>(ws_take(ctx, 16));
        SolveParams sp{p, bias, PDS_SOLVER_QR, prm->l2_reg > 0.0 ? prm->l2_reg : 0.0, 0.0, 0};
        if (int rc = launch_solve<T>(ctx, d_mom, 1, sp, d_beta, d_flag, d_inv, nullptr)) return rc;
        PDS_HIP_CHECK(hipMemcpyAsync(coeffs, d_beta, sizeof(T) * pp, hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipMemcpyAsync(inv, d_inv, sizeof(T) * pp * pp, hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (is_null) *is_null = 0;
        return PDS_OK;
    }
    int null_flag = 0;
    if (int rc = lr_from_device_moments<T>(ctx, d_mom, p, prm, false, coeffs, &null_flag, (T*)nullptr, false, mode == 1)) return rc;
    if (is_null) *is_null = null_flag;
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

}  // namespace pds

// =============================================================================================
// extern "C"
// =============================================================================================
using namespace pds;

extern "C" {

const char* pds_last_error(void) { return g_err.c_str(); }
const char* pds_version(void) { return "pds_lstsq_hip 0.1.0 (gfx950)"; }

int pds_ctx_create(int device, pds_ctx** out) {
    if (!out) return fail(PDS_ERR_INVALID, "null out pointer");
    int ndev = 0;
    PDS_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(PDS_ERR_HIP, "no such HIP device");
    hipDeviceProp_t prop;
    PDS_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(PDS_ERR_HIP, std::string("libpds_lstsq_hip is built for gfx950 only, device is ") + prop.gcnArchName);
    PDS_HIP_CHECK(hipSetDevice(device));
    pds_ctx* c = new pds_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(PDS_ERR_HIP, "hipStreamCreate failed");
    }
    c->own_stream = true;
    const size_t pbytes = (size_t)c->num_cus * 8 * kPartStride * sizeof(double);
    if (hipMalloc(reinterpret_cast<void**>(&c->partials), pbytes) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(PDS_ERR_HIP, "hipMalloc(partials) failed");
    }
    *out = c;
    return PDS_OK;
}

void pds_ctx_destroy(pds_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->partials) (void)hipFree(ctx->partials);
    if (ctx->ws.ptr) (void)hipFree(ctx->ws.ptr);
    for (void* q : ctx->ws_spill) (void)hipFree(q);
    if (ctx->stage.ptr) (void)hipFree(ctx->stage.ptr);
    if (ctx->solve_ws.ptr) (void)hipFree(ctx->solve_ws.ptr);
    if (ctx->keyed.ptr) (void)hipFree(ctx->keyed.ptr);
    if (ctx->mark_count) (void)hipFree(ctx->mark_count);
    if (ctx->mark_host) (void)hipHostFree(ctx->mark_host);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->pinned_in) (void)hipHostFree(ctx->pinned_in);
    for (auto& e : ctx->ev_pending) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int pds_ctx_set_stream(pds_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(PDS_ERR_INVALID, "null ctx");
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    // NULL is a valid hipStream_t: the (legacy) default stream, which is what torch uses unless told otherwise
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
    return PDS_OK;
}

int pds_ctx_synchronize(pds_ctx* ctx) {
    if (!ctx) return fail(PDS_ERR_INVALID, "null ctx");
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

int pds_ctx_num_cus(const pds_ctx* ctx) { return ctx ? ctx->num_cus : 0; }

int pds_glm_irls_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows, pds_space space, int add_bias, int link,
                     int variance, double tol, int max_iter, double* coeffs, int* n_iter) {
    return pds::glm_irls_impl<double>(ctx, cols, n_feat, n_rows, space, add_bias, link, variance, tol, max_iter, coeffs, n_iter);
}
int pds_glm_irls_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows, pds_space space, int add_bias, int link,
                     int variance, float tol, int max_iter, float* coeffs, int* n_iter) {
    return pds::glm_irls_impl<float>(ctx, cols, n_feat, n_rows, space, add_bias, link, variance, tol, max_iter, coeffs, n_iter);
}

int pds_lr_rowmajor_f64(pds_ctx* ctx, const double* X, int64_t ld, const double* y, int64_t n_rows, int n_feat, pds_space space,
                        const pds_lr_params* prm, int mode, double* coeffs, int* is_null, double* inv) {
    return pds::lr_rowmajor_impl<double>(ctx, X, ld, y, n_rows, n_feat, space, prm, mode, coeffs, is_null, inv);
}
int pds_lr_rowmajor_f32(pds_ctx* ctx, const float* X, int64_t ld, const float* y, int64_t n_rows, int n_feat, pds_space space,
                        const pds_lr_params* prm, int mode, float* coeffs, int* is_null, float* inv) {
    return pds::lr_rowmajor_impl<float>(ctx, X, ld, y, n_rows, n_feat, space, prm, mode, coeffs, is_null, inv);
}

int pds_rows_to_cols_f64(pds_ctx* ctx, const double* X, int64_t ld, int64_t n_rows, int n_cols, pds_space space, double* out_cols,
                         int64_t col_stride) {
    return pds::rows_to_cols_impl<double>(ctx, X, ld, n_rows, n_cols, space, out_cols, col_stride);
}
int pds_rows_to_cols_f32(pds_ctx* ctx, const float* X, int64_t ld, int64_t n_rows, int n_cols, pds_space space, float* out_cols,
                         int64_t col_stride) {
    return pds::rows_to_cols_impl<float>(ctx, X, ld, n_rows, n_cols, space, out_cols, col_stride);
}

int pds_set_host_staging(double chunk_mb, double resident_max_mb) {
    if (chunk_mb > 0.0) pds::g_host_chunk_mb = std::max(chunk_mb, 0.001);
    if (resident_max_mb > 0.0) pds::g_host_resident_mb = std::max(resident_max_mb, 0.001);
    return PDS_OK;
}

long long pds_ctx_workspace_spills(const pds_ctx* ctx) { return ctx ? ctx->ws_spill_count : -1; }

int pds_ctx_set_timing(pds_ctx* ctx, int enable) {
    if (!ctx) return fail(PDS_ERR_INVALID, "null ctx");
    ctx->timing = enable != 0;
    return PDS_OK;
}

int pds_ctx_get_timing(pds_ctx* ctx, double* ms_sum, long long* counts, int n_kinds, int reset) {
    if (!ctx) return fail(PDS_ERR_INVALID, "null ctx");
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (auto& e : ctx->ev_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess && e.kind >= 0 && e.kind < 8) {
            ctx->kind_ms[e.kind] += ms;
            ctx->kind_count[e.kind] += 1;
        }
        ctx->ev_pool.push_back(e.a);
        ctx->ev_pool.push_back(e.b);
    }
    ctx->ev_pending.clear();
    for (int k = 0; k < n_kinds && k < 8; ++k) {
        if (ms_sum) ms_sum[k] = ctx->kind_ms[k];
        if (counts) counts[k] = ctx->kind_count[k];
    }
    if (reset)
        for (int k = 0; k < 8; ++k) {
            ctx->kind_ms[k] = 0;
            ctx->kind_count[k] = 0;
        }
    return PDS_OK;
}

int pds_lr_f64(pds_ctx* ctx, const double* const* cols, const double* weights, int n_feat, int64_t n_rows,
               pds_space space, const pds_lr_params* prm, double* coeffs, int* is_null) {
    return lr_impl<double>(ctx, cols, weights, n_feat, n_rows, space, prm, coeffs, is_null, nullptr, nullptr);
}
int pds_lr_f32(pds_ctx* ctx, const float* const* cols, const float* weights, int n_feat, int64_t n_rows,
               pds_space space, const pds_lr_params* prm, float* coeffs, int* is_null) {
    return lr_impl<float>(ctx, cols, weights, n_feat, n_rows, space, prm, coeffs, is_null, nullptr, nullptr);
}
int pds_lr_pred_f64(pds_ctx* ctx, const double* const* cols, const double* weights, int n_feat, int64_t n_rows,
                    pds_space space, const pds_lr_params* prm, double* coeffs, int* is_null, double* pred,
                    double* resid) {
    if (!pred || !resid) return fail(PDS_ERR_INVALID, "pred / resid buffers required");
    return lr_impl<double>(ctx, cols, weights, n_feat, n_rows, space, prm, coeffs, is_null, pred, resid);
}
int pds_lr_pred_f32(pds_ctx* ctx, const float* const* cols, const float* weights, int n_feat, int64_t n_rows,
                    pds_space space, const pds_lr_params* prm, float* coeffs, int* is_null, float* pred, float* resid) {
    if (!pred || !resid) return fail(PDS_ERR_INVALID, "pred / resid buffers required");
    return lr_impl<float>(ctx, cols, weights, n_feat, n_rows, space, prm, coeffs, is_null, pred, resid);
}

int pds_elastic_net_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows, pds_space space,
                        const pds_lr_params* prm, double* coeffs) {
    return lr_impl<double>(ctx, cols, nullptr, n_feat, n_rows, space, prm, coeffs, nullptr, nullptr, nullptr, true);
}
int pds_elastic_net_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows, pds_space space,
                        const pds_lr_params* prm, float* coeffs) {
    return lr_impl<float>(ctx, cols, nullptr, n_feat, n_rows, space, prm, coeffs, nullptr, nullptr, nullptr, true);
}

int pds_lr_nullable_f64(pds_ctx* ctx, const double* const* cols, const uint8_t* const* validity, const int64_t* bit_offsets,
                        int n_feat, int64_t n_rows, pds_space space, int null_policy, double fill_value,
                        const pds_lr_params* prm, double* coeffs, int* is_null, double* pred, double* resid,
                        uint8_t* row_valid, int64_t* n_used) {
    return lr_nullable_impl<double>(ctx, cols, validity, bit_offsets, n_feat, n_rows, space, null_policy, fill_value, prm,
                                    coeffs, is_null, pred, resid, row_valid, n_used);
}
int pds_lr_nullable_f32(pds_ctx* ctx, const float* const* cols, const uint8_t* const* validity, const int64_t* bit_offsets,
                        int n_feat, int64_t n_rows, pds_space space, int null_policy, float fill_value,
                        const pds_lr_params* prm, float* coeffs, int* is_null, float* pred, float* resid,
                        uint8_t* row_valid, int64_t* n_used) {
    return lr_nullable_impl<float>(ctx, cols, validity, bit_offsets, n_feat, n_rows, space, null_policy, fill_value, prm,
                                   coeffs, is_null, pred, resid, row_valid, n_used);
}

int pds_lr_multi_f64(pds_ctx* ctx, const double* const* cols, int n_targets, int n_feat, int64_t n_rows, pds_space space,
                     int add_bias, double l2_reg, int solver, double singular_x_tol, double* coeffs, int* is_null,
                     double* pred, double* resid) {
    return lr_multi_impl<double>(ctx, cols, n_targets, n_feat, n_rows, space, add_bias, l2_reg, solver, singular_x_tol,
                                 coeffs, is_null, pred, resid);
}
int pds_lr_multi_f32(pds_ctx* ctx, const float* const* cols, int n_targets, int n_feat, int64_t n_rows, pds_space space,
                     int add_bias, float l2_reg, int solver, float singular_x_tol, float* coeffs, int* is_null, float* pred,
                     float* resid) {
    return lr_multi_impl<float>(ctx, cols, n_targets, n_feat, n_rows, space, add_bias, l2_reg, solver, singular_x_tol, coeffs,
                                is_null, pred, resid);
}

int pds_lr_rcond_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows, pds_space space,
                     int add_bias, double l2_reg, double rcond, double* coeffs, double* singular_values) {
    return pds::lr_rcond_impl<double>(ctx, cols, n_feat, n_rows, space, add_bias, l2_reg, rcond, coeffs, singular_values);
}
int pds_lr_rcond_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows, pds_space space,
                     int add_bias, float l2_reg, float rcond, float* coeffs, float* singular_values) {
    return pds::lr_rcond_impl<float>(ctx, cols, n_feat, n_rows, space, add_bias, (double)l2_reg, (double)rcond, coeffs,
                                     singular_values);
}

int pds_lin_reg_report_f64(pds_ctx* ctx, const double* const* cols, const double* weights, int n_feat,
                           int64_t n_rows, pds_space space, int add_bias, int se_type, double y_var,
                           pds_report_f64* out) {
    return report_impl<double, pds_report_f64>(ctx, cols, weights, n_feat, n_rows, space, add_bias, se_type, y_var, out);
}
int pds_lin_reg_report_f32(pds_ctx* ctx, const float* const* cols, const float* weights, int n_feat, int64_t n_rows,
                           pds_space space, int add_bias, int se_type, float y_var, pds_report_f32* out) {
    return report_impl<float, pds_report_f32>(ctx, cols, weights, n_feat, n_rows, space, add_bias, se_type, y_var, out);
}

int pds_report_fit_from_moments_f64(pds_ctx* ctx, const double* moments, int n_feat, int add_bias, double* beta, double* inv) {
    return report_fit_impl<double>(ctx, moments, n_feat, add_bias, beta, inv);
}
int pds_report_fit_from_moments_f32(pds_ctx* ctx, const float* moments, int n_feat, int add_bias, float* beta, float* inv) {
    return report_fit_impl<float>(ctx, moments, n_feat, add_bias, beta, inv);
}
int pds_report_partials_f64(pds_ctx* ctx, const double* const* cols, const double* weights, int n_feat, int64_t n_rows,
                            pds_space space, int add_bias, int se_type, const double* beta, const double* inv, double* partials) {
    return report_partials_impl<double>(ctx, cols, weights, n_feat, n_rows, space, add_bias, se_type, beta, inv, partials);
}
int pds_report_partials_f32(pds_ctx* ctx, const float* const* cols, const float* weights, int n_feat, int64_t n_rows,
                            pds_space space, int add_bias, int se_type, const float* beta, const float* inv, double* partials) {
    return report_partials_impl<float>(ctx, cols, weights, n_feat, n_rows, space, add_bias, se_type, beta, inv, partials);
}
int pds_report_finish_f64(int n_feat, int add_bias, int se_type, int weighted, int64_t n_rows_total, double y_var,
                          const double* beta, const double* inv, const double* partials, pds_report_f64* out) {
    return report_finish_impl<double, pds_report_f64>(n_feat, add_bias, se_type, weighted, n_rows_total, y_var, beta, inv, partials, out);
}
int pds_report_finish_f32(int n_feat, int add_bias, int se_type, int weighted, int64_t n_rows_total, float y_var,
                          const float* beta, const float* inv, const double* partials, pds_report_f32* out) {
    return report_finish_impl<float, pds_report_f32>(n_feat, add_bias, se_type, weighted, n_rows_total, y_var, beta, inv, partials, out);
}

int pds_lin_reg_report_nullable_f64(pds_ctx* ctx, const double* const* cols, const uint8_t* const* validity,
                                    const int64_t* bit_offsets, int n_feat, int64_t n_rows, pds_space space, int null_policy,
                                    double fill_value, int add_bias, int se_type, double y_var, pds_report_f64* out,
                                    int64_t* n_used) {
    return report_impl<double, pds_report_f64>(ctx, cols, nullptr, n_feat, n_rows, space, add_bias, se_type, y_var, out, true,
                                               validity, bit_offsets, null_policy, fill_value, n_used);
}
int pds_lin_reg_report_nullable_f32(pds_ctx* ctx, const float* const* cols, const uint8_t* const* validity,
                                    const int64_t* bit_offsets, int n_feat, int64_t n_rows, pds_space space, int null_policy,
                                    float fill_value, int add_bias, int se_type, float y_var, pds_report_f32* out,
                                    int64_t* n_used) {
    return report_impl<float, pds_report_f32>(ctx, cols, nullptr, n_feat, n_rows, space, add_bias, se_type, y_var, out, true,
                                              validity, bit_offsets, null_policy, fill_value, n_used);
}

int pds_lr_grouped_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows,
                       const int64_t* group_offsets, int64_t n_groups, pds_space space, const pds_lr_params* prm,
                       double* coeffs, uint8_t* is_null) {
    return grouped_impl<double>(ctx, cols, n_feat, n_rows, group_offsets, n_groups, space, prm, coeffs, is_null);
}
int pds_lr_grouped_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows,
                       const int64_t* group_offsets, int64_t n_groups, pds_space space, const pds_lr_params* prm,
                       float* coeffs, uint8_t* is_null) {
    return grouped_impl<float>(ctx, cols, n_feat, n_rows, group_offsets, n_groups, space, prm, coeffs, is_null);
}

int pds_lr_grouped_nullable_f64(pds_ctx* ctx, const double* const* cols, const uint8_t* const* validity,
                                const int64_t* bit_offsets, int n_feat, int64_t n_rows, const int64_t* group_offsets,
                                int64_t n_groups, pds_space space, int null_policy, double fill_value, const pds_lr_params* prm,
                                double* coeffs, uint8_t* is_null) {
    return grouped_impl<double>(ctx, cols, n_feat, n_rows, group_offsets, n_groups, space, prm, coeffs, is_null, true, validity,
                                bit_offsets, null_policy, fill_value);
}
int pds_lr_grouped_nullable_f32(pds_ctx* ctx, const float* const* cols, const uint8_t* const* validity,
                                const int64_t* bit_offsets, int n_feat, int64_t n_rows, const int64_t* group_offsets,
                                int64_t n_groups, pds_space space, int null_policy, float fill_value, const pds_lr_params* prm,
                                float* coeffs, uint8_t* is_null) {
    return grouped_impl<float>(ctx, cols, n_feat, n_rows, group_offsets, n_groups, space, prm, coeffs, is_null, true, validity,
                               bit_offsets, null_policy, fill_value);
}

int pds_lr_with_inv_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows, pds_space space, int add_bias,
                        double lambda, double* coeffs, double* inv) {
    return pds::lr_with_inv_impl<double>(ctx, cols, n_feat, n_rows, space, add_bias, lambda, coeffs, inv);
}
int pds_lr_with_inv_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows, pds_space space, int add_bias,
                        float lambda, float* coeffs, float* inv) {
    return pds::lr_with_inv_impl<float>(ctx, cols, n_feat, n_rows, space, add_bias, (double)lambda, coeffs, inv);
}

int pds_lr_grouped_weighted_f64(pds_ctx* ctx, const double* const* cols, const double* weights, int n_feat, int64_t n_rows,
                                const int64_t* group_offsets, int64_t n_groups, pds_space space, const pds_lr_params* prm,
                                double* coeffs, uint8_t* is_null) {
    return pds::grouped_weighted_impl<double>(ctx, cols, weights, n_feat, n_rows, group_offsets, n_groups, space, prm, coeffs, is_null);
}
int pds_lr_grouped_weighted_f32(pds_ctx* ctx, const float* const* cols, const float* weights, int n_feat, int64_t n_rows,
                                const int64_t* group_offsets, int64_t n_groups, pds_space space, const pds_lr_params* prm,
                                float* coeffs, uint8_t* is_null) {
    return pds::grouped_weighted_impl<float>(ctx, cols, weights, n_feat, n_rows, group_offsets, n_groups, space, prm, coeffs, is_null);
}

int pds_lr_by_key_f64(pds_ctx* ctx, const double* const* cols, const int64_t* keys, int n_feat, int64_t n_rows, pds_space space,
                      const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, double* coeffs, uint8_t* is_null,
                      int64_t* n_groups) {
    return pds::lr_by_key_impl<double>(ctx, cols, keys, n_feat, n_rows, space, prm, max_groups, out_keys, coeffs, is_null, n_groups);
}
int pds_lr_by_key_f32(pds_ctx* ctx, const float* const* cols, const int64_t* keys, int n_feat, int64_t n_rows, pds_space space,
                      const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, float* coeffs, uint8_t* is_null,
                      int64_t* n_groups) {
    return pds::lr_by_key_impl<float>(ctx, cols, keys, n_feat, n_rows, space, prm, max_groups, out_keys, coeffs, is_null, n_groups);
}

int pds_rolling_lr_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows, pds_space space,
                       int add_bias, int64_t window, int64_t min_size, double lambda, double* coeffs, double* pred,
                       uint8_t* valid) {
    return rolling_impl<double>(ctx, cols, n_feat, n_rows, space, add_bias, window, min_size, lambda, false, coeffs, pred, valid);
}
int pds_rolling_lr_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows, pds_space space,
                       int add_bias, int64_t window, int64_t min_size, float lambda, float* coeffs, float* pred,
                       uint8_t* valid) {
    return rolling_impl<float>(ctx, cols, n_feat, n_rows, space, add_bias, window, min_size, lambda, false, coeffs, pred, valid);
}
int pds_recursive_lr_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows, pds_space space,
                         int add_bias, int64_t start_with, double lambda, double* coeffs, double* pred,
                         uint8_t* valid) {
    return rolling_impl<double>(ctx, cols, n_feat, n_rows, space, add_bias, start_with, 0, lambda, true, coeffs, pred, valid);
}
int pds_recursive_lr_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows, pds_space space,
                         int add_bias, int64_t start_with, float lambda, float* coeffs, float* pred, uint8_t* valid) {
    return rolling_impl<float>(ctx, cols, n_feat, n_rows, space, add_bias, start_with, 0, lambda, true, coeffs, pred, valid);
}

int pds_recursive_lr_seeded_f64(pds_ctx* ctx, const double* const* cols, int n_feat, int64_t n_rows, pds_space space,
                                int add_bias, int64_t start_with, double lambda, const double* seed_moments,
                                double* coeffs, double* pred, uint8_t* valid) {
    return rolling_impl<double>(ctx, cols, n_feat, n_rows, space, add_bias, start_with, 0, lambda, true, coeffs, pred, valid,
                                seed_moments);
}
int pds_recursive_lr_seeded_f32(pds_ctx* ctx, const float* const* cols, int n_feat, int64_t n_rows, pds_space space,
                                int add_bias, int64_t start_with, float lambda, const float* seed_moments, float* coeffs,
                                float* pred, uint8_t* valid) {
    return rolling_impl<float>(ctx, cols, n_feat, n_rows, space, add_bias, start_with, 0, lambda, true, coeffs, pred, valid,
                               seed_moments);
}

int pds_moments_f64(pds_ctx* ctx, const double* const* cols, const double* weights, int n_feat, int64_t n_rows,
                    pds_space space, double* moments, pds_space out_space) {
    return moments_impl<double>(ctx, cols, weights, n_feat, n_rows, space, moments, out_space);
}
int pds_moments_f32(pds_ctx* ctx, const float* const* cols, const float* weights, int n_feat, int64_t n_rows,
                    pds_space space, float* moments, pds_space out_space) {
    return moments_impl<float>(ctx, cols, weights, n_feat, n_rows, space, moments, out_space);
}
int pds_lr_from_moments_f64(pds_ctx* ctx, const double* moments, pds_space mom_space, int n_feat,
                            const pds_lr_params* prm, double* coeffs, int* is_null) {
    return from_moments_impl<double>(ctx, moments, mom_space, n_feat, prm, coeffs, is_null);
}
int pds_lr_from_moments_f32(pds_ctx* ctx, const float* moments, pds_space mom_space, int n_feat,
                            const pds_lr_params* prm, float* coeffs, int* is_null) {
    return from_moments_impl<float>(ctx, moments, mom_space, n_feat, prm, coeffs, is_null);
}

}  // extern "C"
