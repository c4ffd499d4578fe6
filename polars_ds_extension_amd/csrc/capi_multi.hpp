// capi_multi.hpp -- ONE host frame, SEVERAL contexts: the key-aware grouped fit of a host-resident frame cut into row slices at
// group boundaries, every slice fitted by one of the caller's contexts on that context's own device and stream.
// Part of the one translation unit capi.hip (included there, inside namespace pds, in dependency order).
//
// Why ("host-resident frame: each GPU pulls its own shard over its own PCIe link"): a Polars plugin lives
// inside the Polars process (`register_plugin_function`, python/polars_ds/_utils.py:28-38), so a `torch.distributed` launcher
// cannot serve `pl_lr_by`; what a Polars user holds is a host frame, and a host frame is PCIe bound (14.5 GB over one link:
// 290 ms against 2.5 ms of kernel time).  Eight devices driven from one process pull eight shards over eight links.  On ONE
// device two contexts do something useful as well: slice k + 1 crosses PCIe on one stream while slice k is fitted and its
// results travel back on the other, so only the last slice's fit + D2H + export sit behind the last column.
//
// Contract of the ORDERED route: keys non-decreasing over the whole frame (a frame sorted by its key: groups are contiguous).  Slice boundaries are
// the row counts n s / S moved forward to the next key change, so no group is split and no cross-slice exchange exists; the
// slices' results land in disjoint pieces of the caller's arrays (slice s starts where the groups of slices < s end: a slice
// learns its number of groups before it fits, publishes it, and waits for the counts in front of it).  Frames whose keys are
// NOT in order (round 4): lr_by_key_multi_unordered below -- row slices in any order, per-context moment tables, one sum.
#pragma once
// (<condition_variable>, <mutex>, <thread> are included by capi.hip: this header sits inside namespace pds)

// which route the last pds_lr_by_key_multi_* call of this thread took (tests: 0 single context, 1 ordered slices, 2 unordered slices)
static thread_local int g_multi_route = 0;

template <typename T>
struct SliceBoard {
    std::mutex m;
    std::condition_variable cv;
    std::vector<int64_t> ng;  // groups of slice s, -1 = not known yet
    bool failed = false;      // some slice gave up: nobody waits any longer
    bool unsorted = false;    // ... because its keys were not in order
    int rc = PDS_OK;
    std::string err;
    int64_t max_groups = 0;
    int pp = 0;
    int64_t* out_keys = nullptr;
    T* coeffs = nullptr;
    uint8_t* is_null = nullptr;
    void fail_with(int code, const std::string& msg) {
        std::lock_guard<std::mutex> g(m);
        if (!failed) {
            failed = true;
            rc = code;
            err = msg;
        }
        cv.notify_all();
    }
};

template <typename T>
struct SlicePlace final : ByKeyPlace<T> {
    SliceBoard<T>* b;
    int s;
    SlicePlace(SliceBoard<T>* board, int slice) : b(board), s(slice) {}
    int at(int64_t n_groups, int64_t** out_keys, T** coeffs, uint8_t** is_null) override {
        std::unique_lock<std::mutex> lk(b->m);
        b->ng[s] = n_groups;
        b->cv.notify_all();
        b->cv.wait(lk, [&] {
            if (b->failed) return true;
            for (int k = 0; k < s; ++k)
                if (b->ng[k] < 0) return false;
            return true;
        });
        if (b->failed) return fail(b->rc ? b->rc : PDS_ERR_INVALID, "sliced fit: another slice failed");
        int64_t first = 0;
        for (int k = 0; k < s; ++k) first += b->ng[k];
        if (first + n_groups > b->max_groups) {
            // not an error of this slice alone: every slice still publishes its count so that the caller learns the total
            return fail(PDS_ERR_INVALID, "more distinct keys than max_groups");
        }
        *out_keys = b->out_keys + first;
        *coeffs = b->coeffs + first * b->pp;
        *is_null = b->is_null ? b->is_null + first : nullptr;
        return PDS_OK;
    }
    void unsorted() override {
        std::lock_guard<std::mutex> g(b->m);
        b->unsorted = true;
        b->failed = true;
        b->cv.notify_all();
    }
};

// (host) is the frame cut at `bounds` consistent with non-decreasing keys?  Exact at the cuts, sampled in between: the device
// checks every slice completely once its keys are there, this only keeps obviously unordered frames off the sliced route
static bool keys_look_ordered(const int64_t* keys, int64_t n, const std::vector<int64_t>& bounds) {
    for (size_t s = 1; s + 1 < bounds.size(); ++s)
        if (keys[bounds[s] - 1] >= keys[bounds[s]]) return false;
    const int64_t probes = 4096, step = std::max<int64_t>(1, n / probes);
    for (int64_t i = 0; i + 1 < n; i += step)
        if (keys[i] > keys[i + 1]) return false;
    return true;
}

// slice boundaries: about n / S rows each (at least kMinSliceRows: a slice costs a few launches and copies), moved forward to the next
// key change so that no group is split; `stagger`: the first slice is half a slice
constexpr int64_t kMinSliceRows = (int64_t)1 << 20;
static std::vector<int64_t> slice_bounds(const int64_t* keys, int64_t n_rows, int S, bool stagger) {
    std::vector<int64_t> bounds = {0};
    for (int s = 1; s < S; ++s) {
        int64_t c = stagger ? (int64_t)((double)n_rows * ((double)s - 0.5) / ((double)S - 0.5)) : n_rows / S * s + std::min<int64_t>(s, n_rows % S);
        if (c <= bounds.back()) continue;
        while (c < n_rows && keys[c] == keys[c - 1]) ++c;
        if (c < n_rows && c > bounds.back()) bounds.push_back(c);
    }
    bounds.push_back(n_rows);
    return bounds;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Keys in ANY row order over several contexts / devices ("hash(key) % R" -- here without moving a row to
// "its" device): a group's fit needs its rows' MOMENTS, and moment tables of row slices are additive.  The frame is cut into one
// row slice per context wherever the cut falls; every context pulls its slice over its own link and runs the partition route's
// histogram / scatter / accumulate on it (keyed_partition.hip, phase 1) against the SAME dense ids (base and range of the whole
// frame's keys: one parallel host pass over the keys); the tables meet on the first context (same device: read in place; another
// device: hipMemcpyPeer + add), which lists the groups and solves them (phase 2 + solve_partition_table).  One exchange step of
// ids x nv doubles per extra context -- 440 MB at 1e6 ids x 8 features against the 1.8 GB of frame a context of eight pulls.
// PDS_ERR_UNSUPPORTED (nothing done): the partition route does not apply (sparse keys, > 16 features, ...): the caller keeps the
// single-context sorting route.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T>
static int lr_by_key_multi_unordered(pds_ctx* const* ctxs, int n_ctx, const T* const* cols, const int64_t* keys, int n_feat, int64_t n_rows,
                                     const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, T* coeffs, uint8_t* is_null,
                                     int64_t* n_groups) {
    if (ctxs[0]->opt_keyed_sort || (prm->solver == PDS_SOLVER_SVD && prm->singular_x_tol > 0.0)) return PDS_ERR_UNSUPPORTED;
    if (n_ctx < 2 || n_feat > 16 || n_rows < ((int64_t)1 << 17) || n_rows >= ((int64_t)1 << 31)) return PDS_ERR_UNSUPPORTED;
    const int S = n_ctx, nc = n_feat + 1, pp = n_feat + (prm->add_bias ? 1 : 0);
    std::vector<int64_t> bounds(S + 1);
    for (int s = 0; s <= S; ++s) bounds[s] = n_rows / S * s + std::min<int64_t>(s, n_rows % S);
    // ---- the key range (host, one thread per slice)
    std::vector<int64_t> kmin(S, std::numeric_limits<int64_t>::max()), kmax(S, std::numeric_limits<int64_t>::min());
    {
        std::vector<std::thread> th;
        auto scan = [&](int s) {
            int64_t lo = std::numeric_limits<int64_t>::max(), hi = std::numeric_limits<int64_t>::min();
            for (int64_t i = bounds[s]; i < bounds[s + 1]; ++i) {
                lo = std::min(lo, keys[i]);
                hi = std::max(hi, keys[i]);
            }
            kmin[s] = lo;
            kmax[s] = hi;
        };
        for (int s = 1; s < S; ++s) th.emplace_back(scan, s);
        scan(0);
        for (auto& t : th) t.join();
    }
    const int64_t lo = *std::min_element(kmin.begin(), kmin.end()), hi = *std::max_element(kmax.begin(), kmax.end());
    const int shift = keyed_partition_shift<T>(n_feat);
    const int64_t wdt = (int64_t)1 << shift;
    const int64_t base = lo - (((lo % wdt) + wdt) % wdt);
    if ((uint64_t)hi - (uint64_t)base >= ((uint64_t)1 << 31)) return PDS_ERR_UNSUPPORTED;
    const uint64_t range = (uint64_t)hi - (uint64_t)base + 1;
    const int64_t buckets = keyed_partition_buckets<T>(n_feat, n_rows, range);
    if (buckets <= 0) return PDS_ERR_UNSUPPORTED;
    for (int s = 0; s < S; ++s)
        if (bounds[s + 1] - bounds[s] < ((int64_t)1 << 16)) return PDS_ERR_UNSUPPORTED;
    const int64_t n_ids = keyed_partition_table_ids<T>(n_feat, buckets);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const int64_t cap = std::min<int64_t>(max_groups, n_rows);
    struct Slot {
        KeyedPartitionState st;
        char* pws = nullptr;
        int64_t* d_base = nullptr;
        int rc = PDS_OK;
        std::string err;
    };
    std::vector<Slot> slot(S);
    bool any_peer = false;
    for (int c = 1; c < S; ++c) any_peer = any_peer || ctxs[c]->device != ctxs[0]->device;
    auto build = [&](int c) {
        Slot& me = slot[c];
        try {
            pds_ctx* ctx = ctxs[c];
            auto run = [&]() -> int {
                PDS_HIP_CHECK(hipSetDevice(ctx->device));
                const int64_t r0 = bounds[c], rows = bounds[c + 1] - r0;
                size_t need = 4096 + up((size_t)rows * 8) + (size_t)nc * up((size_t)rows * sizeof(T)) + up(sizeof(T*) * 18) + 256 +
                              keyed_partition_workspace<T>(n_feat, rows, buckets);
                // (first context: group list, results, and -- only when some context sits on another device -- room for a peer's table,
                //  bounded by the partition workspace of zero rows)
                if (c == 0) need += 2 * up((size_t)(cap + 1) * 8) + up((size_t)cap * pp * sizeof(T)) + up((size_t)cap) +
                                    (any_peer ? keyed_partition_workspace<T>(n_feat, 0, buckets) : 0);
                // S id-indexed tables can be several times the frame: a context that cannot get its workspace hands the call back to the
                // single-context route (which sorts or partitions the whole frame in one workspace) instead of failing it
                if (ctx->keyed.bytes < need) {
                    size_t free_b = 0, total_b = 0;
                    PDS_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
                    if (need > free_b + ctx->keyed.bytes) return PDS_ERR_UNSUPPORTED;
                }
                if (ensure_ws(ctx, ctx->keyed, need) != PDS_OK) {
                    (void)hipGetLastError();
                    return PDS_ERR_UNSUPPORTED;
                }
                char* w = static_cast<char*>(ctx->keyed.ptr);
                auto take = [&](size_t b) { char* r = w; w += up(b); return r; };
                int64_t* d_keys = reinterpret_cast<int64_t*>(take((size_t)rows * 8));
                PDS_HIP_CHECK(hipMemcpyAsync(d_keys, keys + r0, (size_t)rows * 8, hipMemcpyHostToDevice, ctx->stream));
                std::vector<const T*> tbl(18, nullptr);
                for (int k = 0; k < nc; ++k) {  // reference order [y, x1..xp] -> kernel order x_0..x_{p-1}, y
                    T* d = reinterpret_cast<T*>(take((size_t)rows * sizeof(T)));
                    PDS_HIP_CHECK(hipMemcpyAsync(d, cols[k] + r0, (size_t)rows * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
                    if (k == 0) tbl[n_feat] = d;
                    else tbl[k - 1] = d;
                }
                for (int k = nc; k < 18; ++k) tbl[k] = tbl[0];
                const T** d_tbl = reinterpret_cast<const T**>(take(sizeof(T*) * 18));
                PDS_HIP_CHECK(hipMemcpyAsync(d_tbl, tbl.data(), sizeof(T*) * 18, hipMemcpyHostToDevice, ctx->stream));
                me.d_base = reinterpret_cast<int64_t*>(take(256));
                PDS_HIP_CHECK(hipMemcpyAsync(me.d_base, &base, 8, hipMemcpyHostToDevice, ctx->stream));
                me.pws = take(keyed_partition_workspace<T>(n_feat, rows, buckets));
                int64_t ng_unused = 0;
                if (int rc = keyed_partition_build<T>(ctx, d_tbl, d_keys, me.d_base, range, n_feat, rows, buckets, me.pws, cap, nullptr, nullptr,
                                                      &ng_unused, me.st, nullptr, 0u, /*phases*/ 1))
                    return rc;
                PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (tbl and base live on this frame; the table is complete)
                return PDS_OK;
            };
            me.rc = run();
            if (me.rc) me.err = g_err;
        } catch (const std::exception& e) {
            me.rc = PDS_ERR_HIP;
            me.err = std::string("sliced fit: ") + e.what();
        }
    };
    {
        std::vector<std::thread> th;
        for (int c = 1; c < S; ++c) th.emplace_back(build, c);
        build(0);
        for (auto& t : th) t.join();
    }
    for (int c = 0; c < S; ++c)
        if (slot[c].rc == PDS_ERR_UNSUPPORTED) return PDS_ERR_UNSUPPORTED;  // (no room for a context's table: the single-context route)
    for (int c = 0; c < S; ++c)
        if (slot[c].rc) return fail(slot[c].rc, slot[c].err.empty() ? std::string("sliced fit failed") : slot[c].err);
    // ---- the exchange step: every other context's table into the first one's
    pds_ctx* c0 = ctxs[0];
    PDS_HIP_CHECK(hipSetDevice(c0->device));
    const size_t table_bytes = (size_t)n_ids * (size_t)slot[0].st.nvp * 8;
    char* w0 = slot[0].pws + up(keyed_partition_workspace<T>(n_feat, bounds[1] - bounds[0], buckets));
    auto take0 = [&](size_t b) { char* r = w0; w0 += up(b); return r; };
    int64_t* d_unique = reinterpret_cast<int64_t*>(take0((size_t)(cap + 1) * 8));
    int64_t* d_offsets = reinterpret_cast<int64_t*>(take0((size_t)(cap + 1) * 8));
    T* d_co = reinterpret_cast<T*>(take0((size_t)cap * pp * sizeof(T)));
    uint8_t* d_nu = reinterpret_cast<uint8_t*>(take0((size_t)cap));
    double* d_peer = nullptr;
    for (int c = 1; c < S; ++c) {
        const double* other = slot[c].st.table;
        if (ctxs[c]->device != c0->device) {
            if (!d_peer) d_peer = reinterpret_cast<double*>(take0(table_bytes));
            PDS_HIP_CHECK(hipMemcpyPeerAsync(d_peer, c0->device, other, ctxs[c]->device, table_bytes, c0->stream));
            other = d_peer;
        }
        if (int rc = keyed_partition_add_table(c0, slot[0].st, n_ids, other)) return rc;
    }
    // ---- the group list and the fits, on the first context
    int64_t ng = 0;
    KeyedPartitionState st = slot[0].st;
    const int rc2 = keyed_partition_build<T>(c0, nullptr, nullptr, slot[0].d_base, range, n_feat, bounds[1] - bounds[0], buckets, slot[0].pws, cap,
                                             d_unique, d_offsets, &ng, st, nullptr, 0u, /*phases*/ 2, n_rows);
    *n_groups = ng;
    if (rc2) return rc2;
    if (int rc = solve_partition_table<T>(c0, st, n_feat, ng, d_offsets, prm, d_co, d_nu)) return rc;
    PDS_HIP_CHECK(hipMemcpyAsync(coeffs, d_co, (size_t)ng * pp * sizeof(T), hipMemcpyDeviceToHost, c0->stream));
    if (is_null) PDS_HIP_CHECK(hipMemcpyAsync(is_null, d_nu, (size_t)ng, hipMemcpyDeviceToHost, c0->stream));
    PDS_HIP_CHECK(hipMemcpyAsync(out_keys, d_unique, (size_t)ng * 8, hipMemcpyDeviceToHost, c0->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(c0->stream));
    return PDS_OK;
}

template <typename T>
static int lr_by_key_multi_impl(pds_ctx* const* ctxs, int n_ctx, int n_slices, const T* const* cols, const int64_t* keys, int n_feat,
                                int64_t n_rows, const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, T* coeffs,
                                uint8_t* is_null, int64_t* n_groups) {
    if (!ctxs || n_ctx < 1 || !cols || !keys || !prm || !out_keys || !coeffs || !n_groups) return fail(PDS_ERR_INVALID, "null argument");
    for (int c = 0; c < n_ctx; ++c)
        if (!ctxs[c]) return fail(PDS_ERR_INVALID, "null context");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    if (max_groups < 1) return fail(PDS_ERR_INVALID, "max_groups must be positive");
    auto single = [&]() {
        g_multi_route = 0;
        return lr_by_key_impl<T>(ctxs[0], cols, keys, n_feat, n_rows, PDS_HOST, prm, max_groups, out_keys, coeffs, is_null, n_groups);
    };
    g_multi_route = 1;
    auto unordered = [&]() {
        g_multi_route = 2;
        // keys not in order: row slices whatever the order, moment tables summed (above); the sorting route on one context otherwise
        const int rcu = lr_by_key_multi_unordered<T>(ctxs, n_ctx, cols, keys, n_feat, n_rows, prm, max_groups, out_keys, coeffs, is_null, n_groups);
        return rcu == PDS_ERR_UNSUPPORTED ? single() : rcu;
    };
    {
        const int64_t probes = 4096, step = std::max<int64_t>(1, n_rows / probes);
        for (int64_t i = 0; i + 1 < n_rows; i += step)
            if (keys[i] > keys[i + 1]) return unordered();
    }
    // ---- slices: at least kMinSliceRows rows each (a slice costs a few launches and copies), cut at key changes
    int S = n_slices > 0 ? n_slices : 4 * n_ctx;
    S = (int)std::min<int64_t>(S, std::max<int64_t>(1, n_rows / kMinSliceRows));
    if (S <= 1 && n_ctx == 1) return single();
    const std::vector<int64_t> bounds = slice_bounds(keys, n_rows, S, false);
    S = (int)bounds.size() - 1;
    if (S <= 1) return single();
    if (!keys_look_ordered(keys, n_rows, bounds)) return unordered();
    const int nc = n_feat + 1, pp = n_feat + (prm->add_bias ? 1 : 0);
    SliceBoard<T> board;
    board.ng.assign(S, -1);
    board.max_groups = max_groups;
    board.pp = pp;
    board.out_keys = out_keys;
    board.coeffs = coeffs;
    board.is_null = is_null;
    const int workers = std::min(n_ctx, S);
    auto work_slices = [&](int c) {
        for (int s = c; s < S; s += workers) {
            const int64_t r0 = bounds[s], rows = bounds[s + 1] - r0;
            {
                std::unique_lock<std::mutex> g(board.m);
                if (board.failed && board.unsorted) return;
                if (board.failed) {
                    // after an overflow (or another slice's failure) the remaining slices only COUNT their groups, on the host: the
                    // caller learns the total it has to size for without this slice's columns crossing the link for nothing
                    g.unlock();
                    int64_t cnt = rows > 0 ? 1 : 0;
                    for (int64_t i = r0 + 1; i < r0 + rows; ++i) cnt += keys[i] != keys[i - 1];
                    g.lock();
                    if (board.ng[s] < 0) board.ng[s] = cnt;
                    board.cv.notify_all();
                    continue;
                }
            }
            std::vector<const T*> ptrs(nc);
            for (int k = 0; k < nc; ++k) ptrs[k] = cols[k] + r0;
            SlicePlace<T> place(&board, s);
            int64_t ng_s = 0;
            const int rc = lr_by_key_impl<T>(ctxs[c], ptrs.data(), keys + r0, n_feat, rows, PDS_HOST, prm, std::min(max_groups, rows), nullptr,
                                             nullptr, nullptr, &ng_s, nullptr, nullptr, nullptr, nullptr, &place);
            if (rc != PDS_OK) {
                {
                    // a slice that stopped before it could publish its count must not leave the slices behind it waiting
                    std::lock_guard<std::mutex> g(board.m);
                    if (board.ng[s] < 0) board.ng[s] = ng_s > 0 ? ng_s : 0;
                }
                board.fail_with(rc, g_err);  // (g_err is this worker's thread-local message)
            }
        }
    };
    // (an exception in a worker -- bad_alloc from a vector -- must not reach std::terminate inside the host process)
    auto work = [&](int c) {
        try {
            work_slices(c);
        } catch (const std::exception& e) {
            {
                std::lock_guard<std::mutex> g(board.m);
                for (int s = c; s < S; s += workers)
                    if (board.ng[s] < 0) board.ng[s] = 0;
            }
            board.fail_with(PDS_ERR_HIP, std::string("sliced fit: ") + e.what());
        }
    };
    std::vector<std::thread> threads;
    for (int c = 1; c < workers; ++c) threads.emplace_back(work, c);
    work(0);
    for (auto& t : threads) t.join();
    if (board.unsorted) return unordered();  // (sampled as ordered, found unordered by a device)
    int64_t total = 0;
    bool all_counted = true;
    for (int s = 0; s < S; ++s) {
        all_counted = all_counted && board.ng[s] >= 0;
        total += std::max<int64_t>(board.ng[s], 0);
    }
    *n_groups = total;
    if (board.failed) {
        if (all_counted && total > max_groups) return fail(PDS_ERR_INVALID, "more distinct keys than max_groups");
        return fail(board.rc ? board.rc : PDS_ERR_INVALID, board.err.empty() ? std::string("sliced fit failed") : board.err);
    }
    return PDS_OK;
}

// ---- per-row predictions of a host frame with non-decreasing keys over several contexts: the slices are fully independent (no
// group list comes back, so there is nothing to place), slice s's pred / resid / row flags land at its own rows.  A slice whose keys
// turn out not to be in order sends the whole frame to the single-context call.
template <typename T>
struct PredSlicePlace final : ByKeyPlace<T> {
    std::atomic<bool>* unsorted_flag;
    explicit PredSlicePlace(std::atomic<bool>* f) : unsorted_flag(f) {}
    int at(int64_t, int64_t**, T**, uint8_t**) override { return PDS_OK; }  // (no coefficient outputs: nothing to place)
    void unsorted() override { unsorted_flag->store(true); }
};

template <typename T>
static int lr_by_key_pred_multi_impl(pds_ctx* const* ctxs, int n_ctx, int n_slices, const T* const* cols, const T* weights, const int64_t* keys,
                                     int n_feat, int64_t n_rows, const pds_lr_params* prm, T* pred, T* resid, uint8_t* row_null) {
    if (!ctxs || n_ctx < 1 || !cols || !keys || !prm || !(pred || resid || row_null)) return fail(PDS_ERR_INVALID, "null argument");
    for (int c = 0; c < n_ctx; ++c)
        if (!ctxs[c]) return fail(PDS_ERR_INVALID, "null context");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    auto single = [&]() {
        return lr_by_key_impl<T>(ctxs[0], cols, keys, n_feat, n_rows, PDS_HOST, prm, n_rows, nullptr, nullptr, nullptr, nullptr, weights, pred,
                                 resid, row_null);
    };
    int S = n_slices > 0 ? n_slices : 4 * n_ctx;
    S = (int)std::min<int64_t>(S, std::max<int64_t>(1, n_rows / kMinSliceRows));
    if (S <= 1) return single();
    // the first slice is HALF a slice: with equal slices the workers run in lockstep -- all of them uploading, then all of them
    // fitting and downloading -- and the predictions' way back (1.7 of 16 GB) never meets an upload; half a slice out of phase, one
    // worker's download runs under the other's upload (the link is full duplex: tools/pcie_duplex.py, 97 GB/s both ways at once)
    const std::vector<int64_t> bounds = slice_bounds(keys, n_rows, S, true);
    S = (int)bounds.size() - 1;
    if (S <= 1 || !keys_look_ordered(keys, n_rows, bounds)) return single();
    const int nc = n_feat + 1;
    std::atomic<bool> unsorted(false), failed(false);
    std::mutex em;
    int first_rc = PDS_OK;
    std::string first_err;
    const int workers = std::min(n_ctx, S);
    auto work = [&](int c) {
        for (int s = c; s < S; s += workers) {
            if (unsorted.load() || failed.load()) return;
            const int64_t r0 = bounds[s], rows = bounds[s + 1] - r0;
            std::vector<const T*> ptrs(nc);
            for (int k = 0; k < nc; ++k) ptrs[k] = cols[k] + r0;
            PredSlicePlace<T> place(&unsorted);
            const int rc = lr_by_key_impl<T>(ctxs[c], ptrs.data(), keys + r0, n_feat, rows, PDS_HOST, prm, rows, nullptr, nullptr, nullptr, nullptr,
                                             weights ? weights + r0 : nullptr, pred ? pred + r0 : nullptr, resid ? resid + r0 : nullptr,
                                             row_null ? row_null + r0 : nullptr, &place);
            if (rc != PDS_OK && !unsorted.load()) {
                std::lock_guard<std::mutex> g(em);
                if (!failed.exchange(true)) {
                    first_rc = rc;
                    first_err = g_err;  // (this worker's thread-local message)
                }
                return;
            }
        }
    };
    std::vector<std::thread> threads;
    for (int c = 1; c < workers; ++c) threads.emplace_back(work, c);
    work(0);
    for (auto& t : threads) t.join();
    if (unsorted.load()) return single();
    if (failed.load()) return fail(first_rc, first_err.empty() ? std::string("sliced fit failed") : first_err);
    return PDS_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// The exchange steps of SURVEY.md 8(e) between the contexts of ONE process (include/pds_lstsq.h): peer copies + one kernel.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void exchange_add_kernel(T* __restrict__ dst, const T* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = (T)((double)dst[i] + (double)src[i]);
}
template <typename T>
static int exchange_copy(pds_ctx* dst_ctx, T* dst, pds_ctx* src_ctx, const T* src, int64_t count) {
    if (count <= 0) return PDS_OK;
    if (dst_ctx->device == src_ctx->device)
        PDS_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)count * sizeof(T), hipMemcpyDeviceToDevice, dst_ctx->stream));
    else
        PDS_HIP_CHECK(hipMemcpyPeerAsync(dst, dst_ctx->device, src, src_ctx->device, (size_t)count * sizeof(T), dst_ctx->stream));
    return PDS_OK;
}
static int exchange_sync_all(pds_ctx* const* ctxs, int n_ctx) {
    for (int c = 0; c < n_ctx; ++c) {
        PDS_HIP_CHECK(hipSetDevice(ctxs[c]->device));
        PDS_HIP_CHECK(hipStreamSynchronize(ctxs[c]->stream));
    }
    return PDS_OK;
}
static int exchange_check(pds_ctx* const* ctxs, int n_ctx) {
    if (!ctxs || n_ctx < 1) return fail(PDS_ERR_INVALID, "null argument");
    for (int c = 0; c < n_ctx; ++c)
        if (!ctxs[c]) return fail(PDS_ERR_INVALID, "null context");
    return PDS_OK;
}

template <typename T>
static int allreduce_sum_impl(pds_ctx* const* ctxs, int n_ctx, T* const* bufs, int64_t count, int prefix) {
    if (int rc = exchange_check(ctxs, n_ctx)) return rc;
    if (!bufs || count < 0) return fail(PDS_ERR_INVALID, "null argument");
    for (int c = 0; c < n_ctx; ++c)
        if (!bufs[c]) return fail(PDS_ERR_INVALID, "null buffer");
    if (count == 0 || (n_ctx == 1 && !prefix)) return PDS_OK;
    if (int rc = exchange_sync_all(ctxs, n_ctx)) return rc;  // (the buffers' producers have finished)
    pds_ctx* c0 = ctxs[0];
    PDS_HIP_CHECK(hipSetDevice(c0->device));
    // everything meets on the first context: acc = running sum in rank order, tmp = the next rank's buffer
    if (int rc = ws_reserve(c0, 2 * ((size_t)count * sizeof(T) + 256) + 4096)) return rc;
    T* acc = reinterpret_cast<T*>(ws_take(c0, (size_t)count * sizeof(T)));
    T* tmp = reinterpret_cast<T*>(ws_take(c0, (size_t)count * sizeof(T)));
    if (!acc || !tmp) return fail(PDS_ERR_HIP, "workspace allocation failed");
    const int nb = (int)std::min<int64_t>((count + 255) / 256, (int64_t)c0->num_cus * 8);
    if (prefix) {
        // exclusive scan: buffer c <- sum of the ORIGINAL buffers 0 .. c-1
        PDS_HIP_CHECK(hipMemsetAsync(acc, 0, (size_t)count * sizeof(T), c0->stream));
        for (int c = 0; c < n_ctx; ++c) {  // (all on the first context's stream: in order)
            if (int rc = exchange_copy<T>(c0, tmp, ctxs[c], bufs[c], count)) return rc;  // the rank's own block
            if (ctxs[c]->device == c0->device)
                PDS_HIP_CHECK(hipMemcpyAsync(bufs[c], acc, (size_t)count * sizeof(T), hipMemcpyDeviceToDevice, c0->stream));
            else
                PDS_HIP_CHECK(hipMemcpyPeerAsync(bufs[c], ctxs[c]->device, acc, c0->device, (size_t)count * sizeof(T), c0->stream));
            hipLaunchKernelGGL((exchange_add_kernel<T>), dim3(nb), dim3(256), 0, c0->stream, acc, (const T*)tmp, count);
        }
        PDS_HIP_CHECK(hipGetLastError());
        return exchange_sync_all(ctxs, n_ctx);
    }
    if (int rc = exchange_copy<T>(c0, acc, c0, bufs[0], count)) return rc;
    for (int c = 1; c < n_ctx; ++c) {
        if (int rc = exchange_copy<T>(c0, tmp, ctxs[c], bufs[c], count)) return rc;
        hipLaunchKernelGGL((exchange_add_kernel<T>), dim3(nb), dim3(256), 0, c0->stream, acc, (const T*)tmp, count);
    }
    PDS_HIP_CHECK(hipGetLastError());
    for (int c = 0; c < n_ctx; ++c) {
        if (ctxs[c]->device == c0->device) PDS_HIP_CHECK(hipMemcpyAsync(bufs[c], acc, (size_t)count * sizeof(T), hipMemcpyDeviceToDevice, c0->stream));
        else PDS_HIP_CHECK(hipMemcpyPeerAsync(bufs[c], ctxs[c]->device, acc, c0->device, (size_t)count * sizeof(T), c0->stream));
    }
    return exchange_sync_all(ctxs, n_ctx);
}

template <typename T>
static int scatter_rows_impl(pds_ctx* const* ctxs, int n_ctx, const T* const* cols, int n_cols, const int64_t* bounds, T* const* const* dst) {
    if (int rc = exchange_check(ctxs, n_ctx)) return rc;
    if (!cols || !bounds || !dst || n_cols < 1) return fail(PDS_ERR_INVALID, "null argument");
    for (int c = 0; c < n_ctx; ++c)
        if (bounds[c + 1] < bounds[c]) return fail(PDS_ERR_INVALID, "bounds must be non-decreasing");
    if (int rc = exchange_sync_all(ctxs, 1)) return rc;
    for (int c = 0; c < n_ctx; ++c) {
        if (!dst[c]) {
            if (c == 0) continue;
            return fail(PDS_ERR_INVALID, "null destination table");
        }
        PDS_HIP_CHECK(hipSetDevice(ctxs[c]->device));  // the receiving rank's stream pulls its shard: all links at once
        for (int k = 0; k < n_cols; ++k) {
            if (!cols[k] || !dst[c][k]) return fail(PDS_ERR_INVALID, "null column");
            if (int rc = exchange_copy<T>(ctxs[c], dst[c][k], ctxs[0], cols[k] + bounds[c], bounds[c + 1] - bounds[c])) return rc;
        }
    }
    return exchange_sync_all(ctxs, n_ctx);
}

template <typename T>
static int gather_impl(pds_ctx* const* ctxs, int n_ctx, const T* const* src, const int64_t* counts, T* dst) {
    if (int rc = exchange_check(ctxs, n_ctx)) return rc;
    if (!src || !counts || !dst) return fail(PDS_ERR_INVALID, "null argument");
    if (int rc = exchange_sync_all(ctxs, n_ctx)) return rc;
    pds_ctx* c0 = ctxs[0];
    PDS_HIP_CHECK(hipSetDevice(c0->device));
    int64_t at = 0;
    for (int c = 0; c < n_ctx; ++c) {
        if (counts[c] < 0 || (counts[c] > 0 && !src[c])) return fail(PDS_ERR_INVALID, "null block");
        if (int rc = exchange_copy<T>(c0, dst + at, ctxs[c], src[c], counts[c])) return rc;
        at += counts[c];
    }
    PDS_HIP_CHECK(hipStreamSynchronize(c0->stream));
    return PDS_OK;
}
