// capi_lr.hpp -- single regressions: pl_lr / pl_lr_pred / nullable / multi-target / rcond / moments (lr_solvers.rs dispatch)
// Part of the one translation unit capi.hip (included there, inside namespace pds, in dependency order): the entry-point
// pipelines are templates with internal linkage, split by concern, not by compilation unit.
#pragma once

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

// The reference's SVD route on one moment record (faer_solve_lr_gated with thin_svd, lr_solvers.rs:358-366; solve :284): gate on
// sum ln sigma_i - sum ln G_ii, beta = V (U'c / sigma).  Returns false when the Jacobi iteration failed on an ungated system
// (the caller falls back to QR, :284-287); `null` = gated / rank-deficient.
template <typename T>
static bool host_svd_gated(const std::vector<T>& M, int p, int bias, double l2, double gate_tol, std::vector<double>& beta, bool& null) {
    const int pp = p + bias;
    std::vector<double> G, c, u, s, v;
    host_normal_eq(M, p, bias, l2, G, c);
    const bool gate = gate_tol > 0.0;
    double ln_den = 0.0;
    null = false;
    if (gate)
        for (int i = 0; i < pp; ++i) {
            if (G[i + (size_t)i * pp] <= 0.0) null = true;
            else ln_den += std::log(G[i + (size_t)i * pp]);
        }
    const bool ok = !null && jacobi_svd(G, pp, u, s, v);
    if (gate && !null) {
        if (!ok) null = true;  // "SVD failure -> treat as rank-deficient" lr_solvers.rs:361-362
        else {
            double ln_det = 0.0;
            for (int i = 0; i < pp; ++i) ln_det += std::log(s[i]);
            if (ln_det - ln_den <= std::log(gate_tol)) null = true;
        }
    }
    beta.assign(pp, NAN);
    if (null) return true;
    if (!ok) return false;
    std::vector<double> z(pp);
    for (int i = 0; i < pp; ++i) {
        double acc = 0;
        for (int r = 0; r < pp; ++r) acc += u[r + (size_t)i * pp] * c[r];
        z[i] = acc / s[i];
    }
    for (int r = 0; r < pp; ++r) {
        double acc = 0;
        for (int i = 0; i < pp; ++i) acc += v[r + (size_t)i * pp] * z[i];
        beta[r] = acc;
    }
    return true;
}

// Grouped fits: the records of the systems the streaming kernels marked (next to the rank gate, gated, broken down) through the
// reference's factorisation for `sp.solver` -- the pivoted QR with the log-det gate (solve.hip) for "qr", the SVD gate above for
// "svd" (per-solver gate, tests/test_linear_exprs.py:1326-1340).  d_co [n][p'], d_fl [n].
// The SVD runs on the host and only for SMALL marked sets (kHostSvdMax systems, a few host threads): a chunk in which most systems sit
// next to the gate (groups with barely more rows than columns, duplicated columns, an overflowing marked list) arrives here whole --
// 1e5+ systems -- and takes the device pivoted QR instead: its gate statistic ln det G - sum ln G_ii is the same number (|det| is the
// product of the R diagonal as well as of the singular values), only the factorisation behind beta differs, to rounding.
// The bound is a COST bound (a Jacobi SVD of a q x q block is O(q^3) per sweep; 2048 systems of 65 x 65 are about two seconds of one
// host thread): narrow systems come in much larger sets -- 5.6e5 at 8 features -- before the route changes.  Beyond it the result
// depends on how many systems a chunk marked (documented: include/pds_lstsq.h "solver").
constexpr int64_t kHostSvdMax = 2048;  // ... systems of the widest LDS-solved block (q = 66)
static inline int64_t host_svd_max_systems(int q) {
    const double unit = 66.0 * 66.0 * 66.0 * (double)kHostSvdMax;
    const double n = unit / ((double)q * q * q);
    return (int64_t)std::min<double>(std::max<double>(n, (double)kHostSvdMax), (double)((int64_t)1 << 20));
}
template <typename T>
int launch_solve_marked(pds_ctx* ctx, const T* d_rec, int64_t n, const SolveParams& sp, T* d_co, uint8_t* d_fl, const int64_t* d_rows_per_sys) {
    if (n <= 0) return PDS_OK;
    SolveParams sq = sp;
    if (sp.solver != PDS_SOLVER_SVD || !(sp.gate_tol > 0.0) || n > host_svd_max_systems(sp.p + 2)) {
        sq.solver = PDS_SOLVER_QR;
        return launch_solve<T>(ctx, d_rec, n, sq, d_co, d_fl, nullptr, d_rows_per_sys);
    }
    const int p = sp.p, bias = sp.add_bias ? 1 : 0, pp = p + bias, q = p + 2;
    std::vector<T> rec, co;
    std::vector<uint8_t> fl;
    std::vector<int64_t> rows;
    try {
        rec.resize((size_t)n * q * q);
        co.resize((size_t)n * pp);
        fl.assign((size_t)n, 0);
        if (d_rows_per_sys) rows.resize((size_t)n + 1);
    } catch (const std::bad_alloc&) {
        return fail(PDS_ERR_HIP, "host staging of the marked systems: out of memory");
    }
    PDS_HIP_CHECK(hipMemcpyAsync(rec.data(), d_rec, rec.size() * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    if (d_rows_per_sys) PDS_HIP_CHECK(hipMemcpyAsync(rows.data(), d_rows_per_sys, rows.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    auto work = [&](int64_t k0, int64_t k1) {
        std::vector<T> M((size_t)q * q);
        std::vector<double> beta;
        for (int64_t k = k0; k < k1; ++k) {
            std::copy(rec.begin() + k * q * q, rec.begin() + (k + 1) * q * q, M.begin());
            bool null = false;
            if (!rows.empty() && rows[k + 1] - rows[k] < pp) null = true, beta.assign(pp, NAN);  // "#Data < #features"
            else (void)host_svd_gated(M, p, bias, sp.lambda, sp.gate_tol, beta, null);  // (gated: a failed iteration is "rank-deficient", never a retry)
            fl[k] = null ? 1 : 0;
            for (int r = 0; r < pp; ++r) co[(size_t)k * pp + r] = (T)beta[r];
        }
    };
    // (a 65 x 65 Jacobi SVD is about a millisecond: beyond a few dozen systems the set is split over up to 16 host threads)
    const int nt = (int)std::min<int64_t>(std::min<int64_t>(16, std::max(1u, std::thread::hardware_concurrency())), (n + 31) / 32);
    if (nt <= 1) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        int64_t done = 0;  // (a host that refuses more threads: the calling thread takes what is left)
        try {
            th.reserve((size_t)nt);
            for (int t = 0; t < nt; ++t) {
                th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
                done = n * (t + 1) / nt;
            }
        } catch (const std::exception&) {
        }
        if (done < n) work(done, n);
        for (auto& t : th) t.join();
    }
    PDS_HIP_CHECK(hipMemcpyAsync(d_co, co.data(), co.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    PDS_HIP_CHECK(hipMemcpyAsync(d_fl, fl.data(), fl.size(), hipMemcpyHostToDevice, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (the staging vectors live on this frame)
    return PDS_OK;
}
template int launch_solve_marked<double>(pds_ctx*, const double*, int64_t, const SolveParams&, double*, uint8_t*, const int64_t*);
template int launch_solve_marked<float>(pds_ctx*, const float*, int64_t, const SolveParams&, float*, uint8_t*, const int64_t*);

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
        std::vector<double> beta;
        bool null = false;
        const bool ok = host_svd_gated(M, p, bias, weighted ? 0.0 : m.l2, weighted ? 0.0 : prm->singular_x_tol, beta, null);
        if (null) {
            for (int i = 0; i < pp; ++i) coeffs[i] = (T)NAN;
            if (is_null) *is_null = 1;
        } else if (ok) {
            for (int r = 0; r < pp; ++r) coeffs[r] = (T)beta[r];
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
    StagingScope staging;  // one snapshot of the host-staging settings for the whole call
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
    StagingScope staging;  // one snapshot of the host-staging settings for the whole call
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
