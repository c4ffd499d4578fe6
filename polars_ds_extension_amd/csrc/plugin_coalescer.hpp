// plugin_coalescer.hpp -- group-commit queue for the per-group pl_lr calls of an unchanged group_by().agg()
// Part of the one translation unit plugin.cpp (included there, inside its anonymous namespace, in dependency order).
#pragma once

// ------------------------------------------------------------------------------------------------- coalescing of per-group calls
// An unchanged `df.group_by(key).agg(pds.lin_reg(...))` makes Polars call `pl_lr` once per group from its rayon threads
// (SURVEY.md 3.3 / 8f rank 1): a million 7 KB frames, each worth two copies, three kernels and a stream synchronisation.
// Calls that arrive while a batch is on the device are queued and leave together as ONE grouped launch (group commit:
// no timer, a lone caller is never held back): the thread that finds nobody serving becomes the server, takes everything
// that is pending with the same parameters, concatenates the frames, runs pds_lr_grouped_* and hands the rows of the
// result back; it serves until the queue is empty.  What takes this route (small frames only): coefficient fits and
// `return_pred` fits (pds_lr_grouped_pred_*: the batch's pred / resid come back as one block and are cut per request), with or
// without weights (pds_lr_grouped_weighted_* / the weights argument of the pred entry point), and coefficient fits on frames
// with nulls under the skip / fill policies (pds_lr_grouped_nullable_*: the requests' validity bitmaps are concatenated bit
// by bit).  Solver "svd" does not (the grouped entry points have no SVD: those calls keep the single-frame path), nor does a
// pred fit on a frame with nulls.  A batch of one runs the same grouped entry point as a batch of a thousand.
// PDS_PLUGIN_COALESCE=0 switches the queue off.
template <typename T>
struct LrRequest {
    const std::vector<Column<T>>* cols;  // [weights?, y, x1..xp], all of length n
    size_t first = 0;                    // index of y in `cols` (1 with weights)
    int n_feat;
    int64_t n;
    pds_lr_params prm;
    bool weighted = false;               // cols[0] = weights: per request faer_weighted_lr
    bool want_pred = false;              // pred / resid (n values each) wanted
    int null_code = 0;                   // 0: null-free frame; PDS_NULL_SKIP / PDS_NULL_FILL: bitmaps travel with the frame
    T fill = T(0);
    std::vector<T>* coeffs;              // out: p' values
    T* pred = nullptr;                   // out (want_pred): n values each
    T* resid = nullptr;
    int is_null = 0;
    std::string error;
    // completion is signalled per request: one shared condition variable woke every waiting thread after every batch
    // (256 rayon threads: 1.7 ms per batch spent in the herd)
    // Completion is handed back as a binary tree of wake-ups: the server marks every request of the batch done and wakes
    // one waiter, every waiter wakes two more before it returns.  (One shared condition variable woke all 256 rayon
    // threads after every batch -- 1.7 ms per batch in the herd; one futex wake per request from the server was the
    // longest part of a 250-request batch; spinning waiters starved the server.)
    // A request may be destroyed as soon as its owner returns, so it returns only after it has been handed the token under
    // its own mutex (notify while holding it), and nobody touches a request after handing it the token.
    std::mutex m;
    std::condition_variable cv;
    bool token = false;
    bool token_handed = false;  // server-side bookkeeping (only the serving thread reads / writes it)
    std::shared_ptr<std::vector<LrRequest*>> wake;  // the batch's waiters (set by the server before the first token)
    size_t wake_idx = 0;

    static void hand_token(LrRequest* q) {
        std::lock_guard<std::mutex> g(q->m);
        q->token = true;
        q->cv.notify_one();
    }
    void wait_done() {
        {
            std::unique_lock<std::mutex> rl(m);
            cv.wait(rl, [&] { return token; });
        }
        const auto w = wake;
        if (w)
            for (size_t c = 2 * wake_idx + 1; c <= 2 * wake_idx + 2 && c < w->size(); ++c) hand_token((*w)[c]);
    }
};

inline bool same_params(const pds_lr_params& a, const pds_lr_params& b) {
    return a.add_bias == b.add_bias && a.l1_reg == b.l1_reg && a.l2_reg == b.l2_reg && a.tol == b.tol && a.solver == b.solver &&
           a.positive == b.positive && a.max_iter == b.max_iter && a.singular_x_tol == b.singular_x_tol;
}

std::atomic<long long> g_coalesce_batches{0}, g_coalesce_requests{0}, g_coalesce_max_batch{0};

template <typename T>
class LrCoalescer {
  public:
    static LrCoalescer& instance() {
        static LrCoalescer q;
        return q;
    }
    // blocks until `r` has its coefficients (or an error message)
    void submit(LrRequest<T>* r) {
        std::unique_lock<std::mutex> lk(m_);
        pending_.push_back(r);  // (a throw here leaves the queue as it was: the caller sees the exception, nobody waits)
        if (serving_) {
            // the server only leaves on an empty queue, and it looks at the queue under this lock: r will be served
            lk.unlock();
            r->wait_done();
            return;
        }
        serving_ = true;
        std::vector<LrRequest<T>*> batch, rest;
        bool batch_done = true;
        // Whatever leaves this scope abnormally (bad_alloc while a batch is put together or its wake-up list is built) must
        // not leave `serving_` set -- every later pl_lr caller would sleep in wait_done for ever -- nor requests that were
        // already taken off the queue without their token: the guard hands every one of them its token, with an error
        // unless its batch had completed, and reopens the queue.
        struct Guard {
            LrCoalescer* q;
            std::unique_lock<std::mutex>& lk;
            LrRequest<T>* self;
            std::vector<LrRequest<T>*>& batch;
            bool& batch_done;
            bool armed = true;
            ~Guard() {
                if (!armed) return;
                if (!lk.owns_lock()) lk.lock();
                std::vector<LrRequest<T>*> orphans;
                orphans.swap(q->pending_);
                q->serving_ = false;
                lk.unlock();
                for (auto* o : batch)
                    if (o != self && !o->token_handed) {
                        if (!batch_done) o->error = "pl_lr: the coalescing queue failed while this request was being served";
                        o->token_handed = true;
                        LrRequest<T>::hand_token(o);
                    }
                for (auto* o : orphans)
                    if (o != self) {
                        o->error = "pl_lr: the coalescing queue failed before this request was served";
                        LrRequest<T>::hand_token(o);
                    }
            }
        } guard{this, lk, r, batch, batch_done};
        while (!pending_.empty()) {
            batch.clear();
            rest.clear();
            batch.push_back(pending_.front());
            for (size_t i = 1; i < pending_.size(); ++i) {
                LrRequest<T>* q = pending_[i];
                if (q->n_feat == batch[0]->n_feat && same_params(q->prm, batch[0]->prm) && q->weighted == batch[0]->weighted &&
                    q->want_pred == batch[0]->want_pred && q->null_code == batch[0]->null_code && q->fill == batch[0]->fill &&
                    (int64_t)batch.size() < kMaxBatch)
                    batch.push_back(q);
                else
                    rest.push_back(q);
            }
            pending_.swap(rest);
            batch_done = false;
            lk.unlock();
            run(batch);  // (catches everything the device path throws and turns it into per-request errors)
            batch_done = true;
            auto waiters = std::make_shared<std::vector<LrRequest<T>*>>();
            for (auto* q : batch)
                if (q != r) waiters->push_back(q);  // (the server's own request needs no wake-up)
            for (size_t i = 0; i < waiters->size(); ++i) {
                (*waiters)[i]->wake = waiters;
                (*waiters)[i]->wake_idx = i;
            }
            // from here on the tree of wake-ups owns the batch: the guard must not touch these requests again
            for (auto* q : *waiters) q->token_handed = true;
            if (!waiters->empty()) LrRequest<T>::hand_token((*waiters)[0]);
            batch.clear();
            lk.lock();
        }
        serving_ = false;
        guard.armed = false;
    }

    static void run_one(LrRequest<T>* r) {  // development: the eligible path without the queue
        std::vector<LrRequest<T>*> b{r};
        run(b);
    }

  private:
    static constexpr int64_t kMaxBatch = 4096;
    std::mutex m_;
    std::vector<LrRequest<T>*> pending_;
    bool serving_ = false;

    static void run(std::vector<LrRequest<T>*>& batch) {
        g_coalesce_batches += 1;
        g_coalesce_requests += (long long)batch.size();
        long long prev = g_coalesce_max_batch.load();
        while ((long long)batch.size() > prev && !g_coalesce_max_batch.compare_exchange_weak(prev, (long long)batch.size())) {
        }
        try {
            const LrRequest<T>& h = *batch[0];
            const int n_feat = h.n_feat, nc = n_feat + 1 + (h.weighted ? 1 : 0);  // [y, x1..xp, (w)]
            const pds_lr_params prm = h.prm;
            const int pp = n_feat + prm.add_bias;
            // (a batch of one takes the grouped entry point as well: which solver answers -- and therefore the null decision
            //  next to singular_x_tol and the coefficients of a rank-deficient group -- must not depend on who else happened
            //  to be queued at that moment)
            int64_t total = 0;
            std::vector<int64_t> off(batch.size() + 1, 0);
            for (size_t g = 0; g < batch.size(); ++g) {
                total += batch[g]->n;
                off[g + 1] = total;
            }
            // (one server at a time: the concatenation buffers are reused across batches -- fresh 50 KB vectors per column
            //  and batch were page faults on the serving thread)
            static std::vector<std::vector<T>> cat;
            if ((int)cat.size() < nc) cat.resize(nc);
            std::vector<const T*> ptrs(nc);
            for (int c = 0; c < nc; ++c) {
                if ((int64_t)cat[c].size() < total) cat[c].resize(total);
                for (size_t g = 0; g < batch.size(); ++g) {
                    const auto& rc = *batch[g]->cols;
                    const Column<T>& src = c < n_feat + 1 ? rc[batch[g]->first + c] : rc[0];  // (the weights are input 0)
                    std::memcpy(cat[c].data() + off[g], src.data(), (size_t)batch[g]->n * sizeof(T));
                }
                ptrs[c] = cat[c].data();
            }
            std::vector<T> co((size_t)batch.size() * pp);
            std::vector<uint8_t> nu(batch.size());
            const T* wts = h.weighted ? ptrs[n_feat + 1] : nullptr;
            if (h.want_pred) {
                static std::vector<T> pr, re;
                if ((int64_t)pr.size() < total) pr.resize(total), re.resize(total);
                check(Api<T>::grouped_pred(thread_ctx(), ptrs.data(), wts, n_feat, total, off.data(), (int64_t)batch.size(), PDS_HOST, &prm,
                                           co.data(), nu.data(), pr.data(), re.data(), nullptr));
                for (size_t g = 0; g < batch.size(); ++g) {
                    std::memcpy(batch[g]->pred, pr.data() + off[g], (size_t)batch[g]->n * sizeof(T));
                    std::memcpy(batch[g]->resid, re.data() + off[g], (size_t)batch[g]->n * sizeof(T));
                }
            } else if (h.weighted) {
                check(Api<T>::grouped_weighted(thread_ctx(), ptrs.data(), wts, n_feat, total, off.data(), (int64_t)batch.size(), PDS_HOST,
                                               &prm, co.data(), nu.data()));
            } else if (h.null_code != 0) {
                // validity of the concatenated frame: bits [off[g], off[g + 1]) of column c = request g's bitmap (all set when the
                // request's column carries none); a column no request has nulls in goes without a bitmap
                std::vector<std::vector<uint8_t>> bms(n_feat + 1);
                std::vector<const uint8_t*> bmp(n_feat + 1, nullptr);
                std::vector<int64_t> boff(n_feat + 1, 0);
                for (int c = 0; c < n_feat + 1; ++c) {
                    bool any = false;
                    for (auto* q : batch) any |= !(*q->cols)[q->first + c].validity.empty();
                    if (!any) continue;
                    bms[c].assign((size_t)(total + 7) / 8, 0);
                    for (size_t g = 0; g < batch.size(); ++g) {
                        const Column<T>& src = (*batch[g]->cols)[batch[g]->first + c];
                        for (int64_t i = 0; i < batch[g]->n; ++i)
                            if (src.validity.empty() || bit_get(src.validity.data(), i)) bit_set(bms[c], off[g] + i);
                    }
                    bmp[c] = bms[c].data();
                }
                check(Api<T>::grouped_nullable(thread_ctx(), ptrs.data(), bmp.data(), boff.data(), n_feat, total, off.data(),
                                               (int64_t)batch.size(), PDS_HOST, h.null_code, h.fill, &prm, co.data(), nu.data()));
            } else {
                check(Api<T>::grouped(thread_ctx(), ptrs.data(), n_feat, total, off.data(), (int64_t)batch.size(), PDS_HOST, &prm, co.data(),
                                      nu.data()));
            }
            for (size_t g = 0; g < batch.size(); ++g) {
                batch[g]->coeffs->assign(co.begin() + g * pp, co.begin() + (g + 1) * pp);
                batch[g]->is_null = nu[g] ? 1 : 0;
            }
        } catch (const PluginError& e) {
            for (auto* q : batch) q->error = e.msg;
        } catch (const std::exception& e) {
            for (auto* q : batch) q->error = e.what();
        }
    }
};

inline bool coalescing_enabled() { return settings().coalesce != 0; }
