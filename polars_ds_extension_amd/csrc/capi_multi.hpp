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
// Contract: keys non-decreasing over the whole frame (a frame sorted by its key: groups are contiguous).  Slice boundaries are
// the row counts n s / S moved forward to the next key change, so no group is split and no cross-slice exchange exists; the
// slices' results land in disjoint pieces of the caller's arrays (slice s starts where the groups of slices < s end: a slice
// learns its number of groups before it fits, publishes it, and waits for the counts in front of it).  Frames whose keys are
// not in order take the single-context route (device sort + gather): distributing THEIR rows by key is a host-side shuffle of
// the whole frame, which costs more than the link saves.
#pragma once
// (<condition_variable>, <mutex>, <thread> are included by capi.hip: this header sits inside namespace pds)

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
        return lr_by_key_impl<T>(ctxs[0], cols, keys, n_feat, n_rows, PDS_HOST, prm, max_groups, out_keys, coeffs, is_null, n_groups);
    };
    // ---- slices: at least kMinSliceRows rows each (a slice costs a few launches and copies), cut at key changes
    int S = n_slices > 0 ? n_slices : 4 * n_ctx;
    S = (int)std::min<int64_t>(S, std::max<int64_t>(1, n_rows / kMinSliceRows));
    if (S <= 1 && n_ctx == 1) return single();
    const std::vector<int64_t> bounds = slice_bounds(keys, n_rows, S, false);
    S = (int)bounds.size() - 1;
    if (S <= 1 || !keys_look_ordered(keys, n_rows, bounds)) return single();
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
    if (board.unsorted) return single();  // (sampled as ordered, found unordered by a device: the whole frame takes the sorting route)
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

