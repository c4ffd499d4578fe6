// plugin_context.hpp -- per-thread pds_ctx and the f64 / f32 entry-point table (Api<T>)
// Part of the one translation unit plugin.cpp (included there, inside its anonymous namespace, in dependency order).
#pragma once

// ------------------------------------------------------------------------------------------------- context
void apply_context_options(pds_ctx* c) {
    (void)pds_ctx_set_option(c, "keyed_sort", settings().keyed_sort ? 1 : 0);
    (void)pds_ctx_set_option(c, "wide_f32_native", settings().wide_f32_native ? 1 : 0);
}
pds_ctx* thread_ctx() {
    // Polars calls plugin symbols from many rayon threads: one context (stream + workspace) per thread
    thread_local pds_ctx* ctx = nullptr;
    thread_local int epoch = 0;  // the settings generation this context's options were taken from
    if (!ctx) {
        if (pds_ctx_create(settings().device, &ctx) != PDS_OK) raise(pds_last_error());
    }
    const int now = g_settings_epoch.load(std::memory_order_acquire);
    if (epoch != now) {
        apply_context_options(ctx);
        epoch = now;
    }
    return ctx;
}
void check(int rc) {
    if (rc != PDS_OK) raise(pds_last_error());
}

// Contexts of the sliced `pl_lr_by` route (pds_lr_by_key_multi_*): PDS_DEVICES = "0,1,..." or "all" names the devices (default:
// the one PDS_DEVICE / 0 names), PDS_BY_KEY_CONTEXTS the contexts per device (default 2: one slice crosses PCIe while the
// previous one is fitted and copied back).  One set per process, used by one call at a time (a second concurrent call finds
// it busy and takes the single-context route on its own thread's context).
struct MultiContexts {
    std::mutex busy;
    std::vector<pds_ctx*> ctxs;
    bool tried = false;
    int epoch = 0;
    static MultiContexts& get() {
        static MultiContexts* m = new MultiContexts();
        return *m;
    }
    // call with `busy` held
    const std::vector<pds_ctx*>& contexts() {
        const int now = g_settings_epoch.load(std::memory_order_acquire);
        if (tried) {
            if (epoch != now)
                for (pds_ctx* c : ctxs) apply_context_options(c);
            epoch = now;
            return ctxs;
        }
        tried = true;
        epoch = now;
        std::vector<int> devs;
        const std::string devices = settings().devices;
        const char* e = devices.c_str();
        if (devices == "all") {
            int n = 0;
            if (pds_device_count(&n) == PDS_OK)
                for (int d = 0; d < n; ++d) devs.push_back(d);
        } else if (*e) {
            std::string tok;
            for (const char* c = e;; ++c) {
                if (*c == ',' || *c == 0) {
                    if (!tok.empty()) devs.push_back(std::atoi(tok.c_str()));
                    tok.clear();
                    if (*c == 0) break;
                } else {
                    tok.push_back(*c);
                }
            }
        }
        if (devs.empty()) devs.push_back(settings().device);
        const int per = settings().by_key_contexts;
        for (int k = 0; k < per; ++k)      // (device-major interleave: slice s goes to device s mod n_dev first)
            for (int d : devs) {
                pds_ctx* c = nullptr;
                if (pds_ctx_create(d, &c) == PDS_OK && c) {
                    apply_context_options(c);
                    ctxs.push_back(c);
                }
            }
        return ctxs;
    }
};
// PDS_REFERENCE_QUIRKS=1 (plugin_settings.hpp): answer the two places where the reference's outputs are accidents of its
// result assembly exactly as it does instead of the way DESIGN.md section 7 argues for --
//   pl_lr_pred, null_policy "ignore", nulls present: ONE row {pred: null, resid: null} (the dummy mask of
//     series_to_mat_for_lr has length 1 and is false, linear_regression.rs:194-197, and :790-806 builds from the mask);
//   pl_recursive_lr, skip / fill, nulls present: pred of the j-th fitted row is formed from compacted row j, not from the
//     row the coefficients belong to (:1158-1166 reads x.get(i..i+1) where the null-free branch reads row m + i).
bool reference_quirks() { return settings().reference_quirks; }

template <typename T> struct Api;
template <> struct Api<double> {
    using Report = pds_report_f64;
    static constexpr auto lr_nullable = pds_lr_nullable_f64;
    static constexpr auto lr = pds_lr_f64;
    static constexpr auto lr_pred = pds_lr_pred_f64;
    static constexpr auto report = pds_lin_reg_report_f64;
    static constexpr auto report_nullable = pds_lin_reg_report_nullable_f64;
    static constexpr auto rolling = pds_rolling_lr_f64;
    static constexpr auto recursive = pds_recursive_lr_f64;
    static constexpr auto by_key = pds_lr_by_key_f64;
    static constexpr auto by_key_pred = pds_lr_by_key_pred_f64;
    static constexpr auto by_key_multi = pds_lr_by_key_multi_f64;
    static constexpr auto by_key_pred_multi = pds_lr_by_key_pred_multi_f64;
    static constexpr auto grouped_pred = pds_lr_grouped_pred_f64;
    static constexpr auto grouped = pds_lr_grouped_f64;
    static constexpr auto grouped_weighted = pds_lr_grouped_weighted_f64;
    static constexpr auto grouped_nullable = pds_lr_grouped_nullable_f64;
    static constexpr auto multi = pds_lr_multi_f64;
    static constexpr auto rcond = pds_lr_rcond_f64;
};
template <> struct Api<float> {
    using Report = pds_report_f32;
    static constexpr auto lr_nullable = pds_lr_nullable_f32;
    static constexpr auto lr = pds_lr_f32;
    static constexpr auto lr_pred = pds_lr_pred_f32;
    static constexpr auto report = pds_lin_reg_report_f32;
    static constexpr auto report_nullable = pds_lin_reg_report_nullable_f32;
    static constexpr auto rolling = pds_rolling_lr_f32;
    static constexpr auto recursive = pds_recursive_lr_f32;
    static constexpr auto by_key = pds_lr_by_key_f32;
    static constexpr auto by_key_pred = pds_lr_by_key_pred_f32;
    static constexpr auto by_key_multi = pds_lr_by_key_multi_f32;
    static constexpr auto by_key_pred_multi = pds_lr_by_key_pred_multi_f32;
    static constexpr auto grouped_pred = pds_lr_grouped_pred_f32;
    static constexpr auto grouped = pds_lr_grouped_f32;
    static constexpr auto grouped_weighted = pds_lr_grouped_weighted_f32;
    static constexpr auto grouped_nullable = pds_lr_grouped_nullable_f32;
    static constexpr auto multi = pds_lr_multi_f32;
    static constexpr auto rcond = pds_lr_rcond_f32;
};

pds_lr_params lr_params(const Kwargs& kw) {  // LRKwargs :27-45 (serde defaults for the optional fields)
    pds_lr_params p;
    p.add_bias = kw_bool(kw, "bias");
    p.l1_reg = kw_f64(kw, "l1_reg");
    p.l2_reg = kw_f64(kw, "l2_reg");
    p.tol = kw_f64(kw, "tol");
    const std::string s = kw_str(kw, "solver", "qr");
    p.solver = s == "svd" ? PDS_SOLVER_SVD : (s == "choleskey" ? PDS_SOLVER_CHOLESKEY : PDS_SOLVER_QR);
    p.positive = kw_bool(kw, "positive");
    p.max_iter = (int)kw_i64(kw, "max_iter");
    p.singular_x_tol = kw_f64(kw, "singular_x_tol");
    return p;
}

template <typename T>
std::vector<Column<T>> import_all(SeriesExport* in, size_t n) {
    std::vector<Column<T>> cols;
    cols.reserve(n);
    for (size_t i = 0; i < n; ++i) cols.push_back(import_series<T>(in[i]));
    return cols;
}
