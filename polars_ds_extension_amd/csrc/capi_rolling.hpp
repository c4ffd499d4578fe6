// capi_rolling.hpp -- rolling / recursive regressions
// Part of the one translation unit capi.hip (included there, inside namespace pds, in dependency order): the entry-point
// pipelines are templates with internal linkage, split by concern, not by compilation unit.
#pragma once

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
