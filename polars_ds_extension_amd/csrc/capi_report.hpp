// capi_report.hpp -- lin_reg_report / wls_report and their row-sharded stages
// Part of the one translation unit capi.hip (included there, inside namespace pds, in dependency order): the entry-point
// pipelines are templates with internal linkage, split by concern, not by compilation unit.
#pragma once

// ---------------------------------------------------------------------------------------------
// lin_reg_report / wls_report
// ---------------------------------------------------------------------------------------------
// second pass over the frame: residuals (sum e^2, sum w e^2) and, for the HC estimators, the per-row weights s_i followed by
// one more *weighted* Gram build = the meat X' diag(s) X (d_mom2: (p+2)^2 moment layout; untouched for plain standard errors)
template <typename T>
static int report_second_pass(pds_ctx* ctx, const DeviceCols<T>& dc, int p, int64_t n_rows, int bias, bool weighted, int se_type,
                              const T* d_beta, const T* d_inv, double* d_sums, T* d_mom2,
                              // PDS_REPORT_DERIVE_YVAR: the pass also sums y - y[0] and its square where it can (<= 16 features) -- d_ysums[0..1],
                              // *ysums_done says whether it did (otherwise the caller runs the separate pass over y)
                              double* d_ysums = nullptr, bool* ysums_done = nullptr) {
    if (ysums_done) *ysums_done = false;
    const int hc = (se_type == PDS_SE) ? 0 : (se_type == PDS_HC2 ? 2 : (se_type == PDS_HC3 ? 3 : 1));
    // HC0 / HC1, p <= 16: the row weights are e_i^2, which the Gram kernel can form itself from the row it has just loaded --
    // residuals, sum e^2 and the meat in ONE pass over the frame (the report is two streams, not three, and the n-row
    // weight vector never exists).  HC2 / HC3 add the leverages (O(p'^2) per row, WM = 4).  Weighted frames and more than 16
    // features: pass2_kernel + a weighted Gram build.
    static const bool no_fuse = [] { const char* e = dev_env("PDS_REPORT_NO_FUSE"); return e && e[0] == '1'; }();
    if (hc == 1 && !weighted && p <= kMaxFeatSmall && !no_fuse) {
        if (ysums_done) *ysums_done = d_ysums != nullptr;
        return launch_moments<T>(ctx, dc, p, n_rows, false, d_mom2, d_beta, bias, d_sums, nullptr, nullptr, d_ysums);
    }
    if (hc >= 2 && !weighted && p <= kMaxFeatSmall && !no_fuse && d_inv) {  // ... and the leverages too: O(p'^2) per row in the same pass
        IrlsArgs ha;
        ha.inv = d_inv;
        ha.hc_pow = hc - 1;
        if (ysums_done) *ysums_done = d_ysums != nullptr;
        return launch_moments<T>(ctx, dc, p, n_rows, false, d_mom2, d_beta, bias, d_sums, nullptr, &ha, d_ysums);
    }
    if constexpr (std::is_same<T, double>::value) {
        // 17 .. 64 f64 features: the same fusion on the streaming multi-tile-column kernel (moments_mid.hip, FUSE)
        if (hc && !weighted && p > kMaxFeatSmall && p <= 64 && !no_fuse && (hc == 1 || d_inv)) {
            const int rc = launch_report_mid(ctx, dc, p, bias, n_rows, d_beta, d_inv, hc, d_sums, d_mom2);
            if (rc != PDS_ERR_UNSUPPORTED) return rc;
        }
    }
    T* d_s = hc ? reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T))) : nullptr;
    const bool ys_here = d_ysums && p <= kMaxFeatSmall;  // (the 17+ feature form of the pass does not carry them)
    if (int rc = launch_pass2<T>(ctx, dc, p, n_rows, bias, weighted, d_beta, d_inv, hc, nullptr, nullptr, d_sums,
                                 reinterpret_cast<double*>(d_s), ys_here ? d_ysums : nullptr))
        return rc;
    if (ysums_done) *ysums_done = ys_here;
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
    const bool derive_var = (se_type & PDS_REPORT_DERIVE_YVAR) != 0 && !weights;  // explicit request, never inferred from y_var
    se_type &= ~PDS_REPORT_DERIVE_YVAR;
    if (se_type < PDS_SE || se_type > PDS_HC3) return fail(PDS_ERR_INVALID, "unknown standard-error type");
    if (weights && se_type != PDS_SE) se_type = PDS_SE;  // pl_wls_report only knows "std_err"
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int p = n_feat, bias = add_bias ? 1 : 0, pp = p + bias, q = p + 2;
    size_t need = 131072 + 65536 + sizeof(T) * (size_t)(2 * q * q + pp * pp + pp) + sizeof(T*) * (size_t)(p + 64);
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
    // PDS_REPORT_DERIVE_YVAR: sums of y - y[0] for var(y) -- from the second pass itself where it reads y anyway (<= 16 features), one
    // more pass over the target column otherwise
    double* d_ys = derive_var ? reinterpret_cast<double*>(ws_take(ctx, 64)) : nullptr;
    bool ys_done = false;
    if (int rc = report_second_pass<T>(ctx, dc, p, n_rows, bias, weighted, se_type, d_beta, d_inv, d_sums, d_mom2, d_ys, &ys_done)) return rc;
    const bool hc = se_type != PDS_SE;
    std::vector<T> beta(pp), inv((size_t)pp * pp), meat((size_t)q * q);
    double sums[2] = {0, 0};
    PDS_HIP_CHECK(hipMemcpyAsync(beta.data(), d_beta, sizeof(T) * pp, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipMemcpyAsync(inv.data(), d_inv, sizeof(T) * pp * pp, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipMemcpyAsync(sums, d_sums, sizeof(sums), hipMemcpyDeviceToHost, ctx->stream));
    if (hc) PDS_HIP_CHECK(hipMemcpyAsync(meat.data(), d_mom2, sizeof(T) * q * q, hipMemcpyDeviceToHost, ctx->stream));
    // PDS_REPORT_DERIVE_YVAR: the target's sample variance (ddof = 1, what Polars evaluates as `target.var()` and hands over as
    // input 0, expr_linear.py:614-617).  f64: sum y and sum y^2 are entries of the moment matrix this call has just built.  f32:
    // those entries are rounded to f32 and syy - sy^2 / n would cancel for |mean| >> std, so y is summed once more in f64.
    double mom_y[2] = {0.0, 0.0};
    if (derive_var) {
        // var(y) as Polars' `target.var()` delivers it to the reference (numerically stable): from the sums of y - y[0], one more
        // pass over the target column -- the raw moments of the Gram record (sum y, sum y^2) cancel when |mean| >> std
        if (!ys_done)
            if (int rc = launch_y_sums<T>(ctx, dc.h_ptrs[p], n_rows, d_ys, true)) return rc;
        PDS_HIP_CHECK(hipMemcpyAsync(mom_y, d_ys, sizeof(mom_y), hipMemcpyDeviceToHost, ctx->stream));
    }
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (derive_var) {
        const double nn = (double)n_rows, sy = mom_y[0], syy = mom_y[1];
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
