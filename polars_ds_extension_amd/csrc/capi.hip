// capi.hip -- the extern "C" surface declared in include/pds_lstsq.h: context management, host <-> HBM
// staging, and the per-expression pipelines that string the kernels together.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include <hipcub/hipcub.hpp>

#include "common.hpp"

namespace pds {

#include "capi_core.hpp"
#include "capi_staging.hpp"
#include "capi_lr.hpp"
#include "capi_report.hpp"
#include "capi_grouped.hpp"
#include "capi_multi.hpp"
#include "capi_rolling.hpp"
#include "capi_models.hpp"

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
    {   // option defaults from the environment, once per context (pds_ctx_set_option changes them afterwards)
        const char* ks = std::getenv("PDS_KEYED_SORT");
        c->opt_keyed_sort = ks && ks[0] == '1';
        const char* wn = std::getenv("PDS_WIDE_F32_NATIVE");
        c->opt_wide_f32_native = wn && wn[0] == '1';
    }
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
    if (ctx->wkeyed.ptr) (void)hipFree(ctx->wkeyed.ptr);
    if (ctx->mark_count) (void)hipFree(ctx->mark_count);
    if (ctx->wait_timeouts) (void)hipFree(ctx->wait_timeouts);
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

int pds_ctx_set_option(pds_ctx* ctx, const char* name, long long value) {
    if (!ctx || !name) return fail(PDS_ERR_INVALID, "null argument");
    const std::string n(name);
    if (n == "keyed_sort") ctx->opt_keyed_sort = value != 0;
    else if (n == "wide_f32_native") ctx->opt_wide_f32_native = value != 0;
    else return fail(PDS_ERR_INVALID, "unknown context option: " + n);
    return PDS_OK;
}

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
    if (chunk_mb > 0.0) pds::g_host_chunk_mb.store(std::max(chunk_mb, 0.001), std::memory_order_relaxed);
    if (resident_max_mb > 0.0) pds::g_host_resident_mb.store(std::max(resident_max_mb, 0.001), std::memory_order_relaxed);
    return PDS_OK;
}

long long pds_ctx_workspace_spills(const pds_ctx* ctx) { return ctx ? ctx->ws_spill_count : -1; }

long long pds_ctx_workspace_bytes(const pds_ctx* ctx, int which) {
    if (!ctx) return -1;
    const pds::Workspace* w[5] = {&ctx->ws, &ctx->stage, &ctx->solve_ws, &ctx->keyed, &ctx->wkeyed};
    return which >= 0 && which < 5 ? (long long)w[which]->bytes : -1;
}

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
            auto& sm = ctx->kind_samples[e.kind];
            if (sm.size() >= 4096) sm.erase(sm.begin(), sm.begin() + 2048);
            sm.push_back(ms);
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

int pds_ctx_get_timing_samples(pds_ctx* ctx, int kind, double* ms_out, int cap, int reset) {
    if (!ctx || kind < 0 || kind >= 8) return -1;
    if (pds_ctx_get_timing(ctx, nullptr, nullptr, 0, 0) != PDS_OK) return -1;  // drains the pending event pairs
    auto& sm = ctx->kind_samples[kind];
    const int n = (int)std::min<size_t>(sm.size(), (size_t)std::max(cap, 0));
    for (int i = 0; i < n; ++i) ms_out[i] = sm[sm.size() - n + i];
    if (reset) sm.clear();
    return n;
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

int pds_lr_by_key_multi_f64(pds_ctx* const* ctxs, int n_ctx, int n_slices, const double* const* cols, const int64_t* keys, int n_feat,
                            int64_t n_rows, const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, double* coeffs,
                            uint8_t* is_null, int64_t* n_groups) {
    return pds::lr_by_key_multi_impl<double>(ctxs, n_ctx, n_slices, cols, keys, n_feat, n_rows, prm, max_groups, out_keys, coeffs, is_null,
                                             n_groups);
}
int pds_lr_by_key_multi_f32(pds_ctx* const* ctxs, int n_ctx, int n_slices, const float* const* cols, const int64_t* keys, int n_feat,
                            int64_t n_rows, const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, float* coeffs,
                            uint8_t* is_null, int64_t* n_groups) {
    return pds::lr_by_key_multi_impl<float>(ctxs, n_ctx, n_slices, cols, keys, n_feat, n_rows, prm, max_groups, out_keys, coeffs, is_null,
                                            n_groups);
}
int pds_lr_by_key_pred_multi_f64(pds_ctx* const* ctxs, int n_ctx, int n_slices, const double* const* cols, const double* weights,
                                 const int64_t* keys, int n_feat, int64_t n_rows, const pds_lr_params* prm, double* pred, double* resid,
                                 uint8_t* row_null) {
    return pds::lr_by_key_pred_multi_impl<double>(ctxs, n_ctx, n_slices, cols, weights, keys, n_feat, n_rows, prm, pred, resid, row_null);
}
int pds_lr_by_key_pred_multi_f32(pds_ctx* const* ctxs, int n_ctx, int n_slices, const float* const* cols, const float* weights,
                                 const int64_t* keys, int n_feat, int64_t n_rows, const pds_lr_params* prm, float* pred, float* resid,
                                 uint8_t* row_null) {
    return pds::lr_by_key_pred_multi_impl<float>(ctxs, n_ctx, n_slices, cols, weights, keys, n_feat, n_rows, prm, pred, resid, row_null);
}
// pinned (page-locked, portable) host storage for results: a device-to-host copy into it runs at the link rate instead of the
// pageable rate (7.8 -> ~2.5 ms for the 136 MB of the headline frame's coefficients)
int pds_host_alloc(size_t bytes, void** out) {
    if (!out) return pds::fail(PDS_ERR_INVALID, "null argument");
    *out = nullptr;
    PDS_HIP_CHECK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable));
    return PDS_OK;
}
int pds_host_free(void* p) {
    if (p) PDS_HIP_CHECK(hipHostFree(p));
    return PDS_OK;
}
int pds_device_count(int* n) {
    if (!n) return pds::fail(PDS_ERR_INVALID, "null argument");
    *n = 0;
    PDS_HIP_CHECK(hipGetDeviceCount(n));
    return PDS_OK;
}
// result storage a peer PROCESS's kernels write into (the direct gather of the group-sharded step: include/pds_lstsq.h)
static_assert(sizeof(hipIpcMemHandle_t) == PDS_IPC_HANDLE_BYTES, "the handle travels as 64 opaque bytes");
int pds_device_alloc(int device, size_t bytes, int fine_grained, void** out) {
    if (!out) return pds::fail(PDS_ERR_INVALID, "null argument");
    *out = nullptr;
    PDS_HIP_CHECK(hipSetDevice(device));
    if (fine_grained) PDS_HIP_CHECK(hipExtMallocWithFlags(out, bytes ? bytes : 1, hipDeviceMallocFinegrained));
    else PDS_HIP_CHECK(hipMalloc(out, bytes ? bytes : 1));
    return PDS_OK;
}
// stream-ordered completion words of the direct gather (include/pds_lstsq.h)
namespace pds {
__global__ void signal_post_kernel(unsigned* word, unsigned value) {
    __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(64) void signal_wait_kernel(const unsigned* words, int n, unsigned value, long long budget_ticks, int* timed_out) {
    const int lane = threadIdx.x;
    const long long t0 = wall_clock64();
    bool ok = lane >= n;
    while (!__all(ok)) {
        if (!ok) ok = (int)(__hip_atomic_load(words + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) >= 0;  // (wrap-safe >=)
        if (wall_clock64() - t0 > budget_ticks) {
            if (lane == 0) atomicAdd(timed_out, 1);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}
}  // namespace pds
int pds_signal_post(pds_ctx* ctx, unsigned* word, unsigned value) {
    if (!ctx || !word) return pds::fail(PDS_ERR_INVALID, "null argument");
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(pds::signal_post_kernel, dim3(1), dim3(1), 0, ctx->stream, word, value);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
int pds_signal_wait(pds_ctx* ctx, const unsigned* words, int n_words, unsigned value, int timeout_ms) {
    if (!ctx || !words || n_words < 0 || n_words > 64) return pds::fail(PDS_ERR_INVALID, "pds_signal_wait: up to 64 words");
    if (n_words == 0) return PDS_OK;
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->wait_timeouts) {
        PDS_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx->wait_timeouts), 64));
        PDS_HIP_CHECK(hipMemsetAsync(ctx->wait_timeouts, 0, 64, ctx->stream));
    }
    int rate_khz = 100000;  // wall_clock64 ticks per ms
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, ctx->device);
    const long long budget = (long long)(timeout_ms > 0 ? timeout_ms : 2000) * (long long)rate_khz;
    hipLaunchKernelGGL(pds::signal_wait_kernel, dim3(1), dim3(64), 0, ctx->stream, words, n_words, value, budget,
                       ctx->wait_timeouts);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
int pds_signal_wait_status(pds_ctx* ctx, int* timed_out) {
    if (!ctx || !timed_out) return pds::fail(PDS_ERR_INVALID, "null argument");
    *timed_out = 0;
    if (!ctx->wait_timeouts) return PDS_OK;
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    PDS_HIP_CHECK(hipMemcpyAsync(timed_out, ctx->wait_timeouts, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}
int pds_device_free(int device, void* p) {
    if (!p) return PDS_OK;
    PDS_HIP_CHECK(hipSetDevice(device));
    PDS_HIP_CHECK(hipFree(p));
    return PDS_OK;
}
int pds_ipc_export(int device, void* p, unsigned char* handle) {
    if (!p || !handle) return pds::fail(PDS_ERR_INVALID, "null argument");
    PDS_HIP_CHECK(hipSetDevice(device));
    hipIpcMemHandle_t h;
    PDS_HIP_CHECK(hipIpcGetMemHandle(&h, p));
    std::memcpy(handle, &h, sizeof(h));
    return PDS_OK;
}
int pds_ipc_open(int device, const unsigned char* handle, void** mapped) {
    if (!handle || !mapped) return pds::fail(PDS_ERR_INVALID, "null argument");
    *mapped = nullptr;
    PDS_HIP_CHECK(hipSetDevice(device));
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof(h));
    PDS_HIP_CHECK(hipIpcOpenMemHandle(mapped, h, hipIpcMemLazyEnablePeerAccess));
    return PDS_OK;
}
int pds_ipc_close(int device, void* mapped) {
    if (!mapped) return PDS_OK;
    PDS_HIP_CHECK(hipSetDevice(device));
    PDS_HIP_CHECK(hipIpcCloseMemHandle(mapped));
    return PDS_OK;
}

int pds_lr_grouped_pred_f64(pds_ctx* ctx, const double* const* cols, const double* weights, int n_feat, int64_t n_rows,
                            const int64_t* group_offsets, int64_t n_groups, pds_space space, const pds_lr_params* prm, double* coeffs,
                            uint8_t* is_null, double* pred, double* resid, uint8_t* row_null) {
    if (!pred && !resid && !row_null) return pds::fail(PDS_ERR_INVALID, "pds_lr_grouped_pred: no per-row output requested");
    if (weights)
        return pds::grouped_weighted_impl<double>(ctx, cols, weights, n_feat, n_rows, group_offsets, n_groups, space, prm, coeffs, is_null,
                                                  pred, resid, row_null, nullptr);
    return pds::grouped_impl<double>(ctx, cols, n_feat, n_rows, group_offsets, n_groups, space, prm, coeffs, is_null, false, nullptr,
                                     nullptr, PDS_NULL_RAISE, 0.0, pred, resid, row_null);
}
int pds_lr_grouped_pred_f32(pds_ctx* ctx, const float* const* cols, const float* weights, int n_feat, int64_t n_rows,
                            const int64_t* group_offsets, int64_t n_groups, pds_space space, const pds_lr_params* prm, float* coeffs,
                            uint8_t* is_null, float* pred, float* resid, uint8_t* row_null) {
    if (!pred && !resid && !row_null) return pds::fail(PDS_ERR_INVALID, "pds_lr_grouped_pred: no per-row output requested");
    if (weights)
        return pds::grouped_weighted_impl<float>(ctx, cols, weights, n_feat, n_rows, group_offsets, n_groups, space, prm, coeffs, is_null,
                                                 pred, resid, row_null, nullptr);
    return pds::grouped_impl<float>(ctx, cols, n_feat, n_rows, group_offsets, n_groups, space, prm, coeffs, is_null, false, nullptr,
                                    nullptr, PDS_NULL_RAISE, 0.0f, pred, resid, row_null);
}
int pds_lr_by_key_pred_f64(pds_ctx* ctx, const double* const* cols, const double* weights, const int64_t* keys, int n_feat,
                           int64_t n_rows, pds_space space, const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys,
                           double* coeffs, uint8_t* is_null, int64_t* n_groups, double* pred, double* resid, uint8_t* row_null) {
    return pds::lr_by_key_impl<double>(ctx, cols, keys, n_feat, n_rows, space, prm, max_groups, out_keys, coeffs, is_null, n_groups, weights,
                                       pred, resid, row_null);
}
int pds_lr_by_key_pred_f32(pds_ctx* ctx, const float* const* cols, const float* weights, const int64_t* keys, int n_feat, int64_t n_rows,
                           pds_space space, const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, float* coeffs,
                           uint8_t* is_null, int64_t* n_groups, float* pred, float* resid, uint8_t* row_null) {
    return pds::lr_by_key_impl<float>(ctx, cols, keys, n_feat, n_rows, space, prm, max_groups, out_keys, coeffs, is_null, n_groups, weights,
                                      pred, resid, row_null);
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

// ---- exchange steps between the contexts of one process (capi_multi.hpp)
int pds_allreduce_sum_f64(pds_ctx* const* ctxs, int n_ctx, double* const* bufs, int64_t count, int prefix) {
    return pds::allreduce_sum_impl<double>(ctxs, n_ctx, bufs, count, prefix);
}
int pds_allreduce_sum_f32(pds_ctx* const* ctxs, int n_ctx, float* const* bufs, int64_t count, int prefix) {
    return pds::allreduce_sum_impl<float>(ctxs, n_ctx, bufs, count, prefix);
}
int pds_scatter_rows_f64(pds_ctx* const* ctxs, int n_ctx, const double* const* cols, int n_cols, const int64_t* bounds, double* const* const* dst) {
    return pds::scatter_rows_impl<double>(ctxs, n_ctx, cols, n_cols, bounds, dst);
}
int pds_scatter_rows_f32(pds_ctx* const* ctxs, int n_ctx, const float* const* cols, int n_cols, const int64_t* bounds, float* const* const* dst) {
    return pds::scatter_rows_impl<float>(ctxs, n_ctx, cols, n_cols, bounds, dst);
}
int pds_gather_f64(pds_ctx* const* ctxs, int n_ctx, const double* const* src, const int64_t* counts, double* dst) {
    return pds::gather_impl<double>(ctxs, n_ctx, src, counts, dst);
}
int pds_gather_f32(pds_ctx* const* ctxs, int n_ctx, const float* const* src, const int64_t* counts, float* dst) {
    return pds::gather_impl<float>(ctxs, n_ctx, src, counts, dst);
}
int pds_debug_last_multi_route(void) { return pds::g_multi_route; }
int pds_debug_last_grouped_route(void) { return pds::g_grouped_route; }
}  // extern "C"
