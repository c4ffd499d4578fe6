// capi_core.hpp -- error state, HIP-event timing, workspaces, host column staging (make_device_cols)
// Part of the one translation unit capi.hip (included there, inside namespace pds, in dependency order): the entry-point
// pipelines are templates with internal linkage, split by concern, not by compilation unit.
#pragma once

static thread_local std::string g_err;

// PDS_TRACE=1: wall-clock stage marks of the host-frame pipelines on stderr (each mark synchronises the stream: a diagnostic,
// it serialises what would otherwise overlap)
struct StageTrace {
    pds_ctx* ctx;
    const char* what;
    bool on;
    std::chrono::steady_clock::time_point t0;
    StageTrace(pds_ctx* c, const char* w) : ctx(c), what(w) {
        static const bool env = [] { const char* e = std::getenv("PDS_TRACE"); return e && e[0] == '1'; }();
        on = env;
        if (on) t0 = std::chrono::steady_clock::now();
    }
    void mark(const char* stage) {
        if (!on) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pds trace] %s: %-28s %8.2f ms\n", what, stage, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
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
