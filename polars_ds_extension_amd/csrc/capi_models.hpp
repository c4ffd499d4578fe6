// capi_models.hpp -- the pyclass route: GLM (IRLS) and fits from row-major matrices
// Part of the one translation unit capi.hip (included there, inside namespace pds, in dependency order): the entry-point
// pipelines are templates with internal linkage, split by concern, not by compilation unit.
#pragma once

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
    if (n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    if (max_iter < 1) return fail(PDS_ERR_INVALID, "`max_iter` must be > 1.");  // linear_models.py:756-757
    if (link < 0 || link > 3 || variance < 0 || variance > 3) return fail(PDS_ERR_INVALID, "unknown link / variance function");
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int p = n_feat, bias = add_bias ? 1 : 0, pp = p + bias, q = p + 2;
    const bool wide = p > kMaxFeatSmall;  // weights / working response as columns of their own + the wide weighted Gram build
    const size_t row_bytes = ((size_t)n_rows * sizeof(T) + 255) & ~(size_t)255;
    if (int rc = ws_reserve(ctx, 262144 + (size_t)max_iter * 1024 + sizeof(T) * (size_t)(2 * q * q + 2 * pp + 16) + sizeof(T*) * (2 * (size_t)p + 64) +
                                     (wide ? 2 * row_bytes + moments_wide_workspace(ctx->num_cus, p, n_rows, true) : 0)))
        return rc;
    DeviceCols<T> dc;
    if (int rc = make_device_cols<T>(ctx, cols, (const T*)nullptr, p, n_rows, space, dc)) return rc;
    T* d_mom = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * q * q));
    T* d_beta = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * (pp + 2)));
    IrlsArgs ia;
    ia.link = link;
    ia.variance = variance;
    ia.init = 1;
    if (variance != 2) {  // mean of y for the starting mu (:272-279): sum(y) is an entry of a plain moment matrix
        T sy = T(0);
        if (!wide) {
            if (int rc = launch_moments<T>(ctx, dc, p, n_rows, false, d_mom)) return rc;
            PDS_HIP_CHECK(hipMemcpyAsync(&sy, d_mom + p + (size_t)(p + 1) * q, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        } else {  // ... of the one-feature frame [y | y]: a streaming pass over y instead of a (p + 2)^2 Gram build
            const T* yy[2] = {dc.h_ptrs[p], dc.h_ptrs[p]};
            DeviceCols<T> dy;
            if (int rc = make_device_cols<T>(ctx, yy, (const T*)nullptr, 1, n_rows, PDS_DEVICE, dy)) return rc;
            if (int rc = launch_moments<T>(ctx, dy, 1, n_rows, false, d_mom)) return rc;
            PDS_HIP_CHECK(hipMemcpyAsync(&sy, d_mom + 1, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));  // Z = [y | 1 | y]: (1, 0)
        }
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ia.y_mean = (double)sy / (double)n_rows;
    }
    DeviceCols<T> dcw;  // wide: [z | x1..xp] with the weight column w
    T *d_w = nullptr, *d_z = nullptr;
    if (wide) {
        d_w = reinterpret_cast<T*>(ws_take(ctx, row_bytes));
        d_z = reinterpret_cast<T*>(ws_take(ctx, row_bytes));
        std::vector<const T*> cw(p + 1);
        cw[0] = d_z;
        for (int c = 0; c < p; ++c) cw[c + 1] = dc.h_ptrs[c];
        if (int rc = make_device_cols<T>(ctx, cw.data(), d_w, p, n_rows, PDS_DEVICE, dcw)) return rc;
    }
    pds_lr_params prm{};
    prm.add_bias = bias;
    prm.solver = PDS_SOLVER_QR;  // GLM::fit_unchecked passes LRSolverMethods::QR (:226, :236)
    prm.max_iter = 1;
    std::vector<T> beta(pp, T(0)), bnew(pp, T(0));
    int it = 0;
    while (it < max_iter) {
        ++it;
        const size_t ws_mark = ctx->ws_used;
        if (!wide) {
            if (int rc = launch_moments<T>(ctx, dc, p, n_rows, false, d_mom, d_beta, bias, nullptr, nullptr, &ia)) return rc;
        } else {
            if (int rc = launch_irls_working_wide<T>(ctx, dc, p, n_rows, bias, d_beta, ia, d_w, d_z)) return rc;
            if (int rc = launch_moments_wide<T>(ctx, dcw, p, n_rows, true, d_mom)) return rc;
        }
        int null_flag = 0;
        if (int rc = lr_from_device_moments<T>(ctx, d_mom, p, &prm, /*weighted=*/true, bnew.data(), &null_flag, d_beta)) return rc;
        if (wide) ctx->ws_used = ws_mark;  // the partial tiles of the wide build are per iteration (the solve has synchronised)
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
