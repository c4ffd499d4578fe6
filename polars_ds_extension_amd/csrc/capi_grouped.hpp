// capi_grouped.hpp -- grouped regressions: contiguous groups, weighted groups, int64 keys in any row order
// Part of the one translation unit capi.hip (included there, inside namespace pds, in dependency order): the entry-point
// pipelines are templates with internal linkage, split by concern, not by compilation unit.
#pragma once

// ---------------------------------------------------------------------------------------------
// grouped
// ---------------------------------------------------------------------------------------------
// which way the last grouped call of this thread with 17 .. 32 f64 features went (tests): 0 neither, 1 the fused stream, 2 the record
// pipeline after the fused form gave the call back (marked list overflow)
static thread_local int g_grouped_route = 0;
// Groups with more than 64 features (coverage path): every group's Gram matrix comes from the tiled matrix-core SYRK of the
// single-regression path (moments_wide.hip) on that group's row range, the records of a chunk of groups are then solved
// together (solve_big.hip: Cholesky on an L2-resident workspace, one workgroup per system; CD / NNLS: one wavefront each).
__global__ void mark_small_groups_kernel(const int64_t* __restrict__ off, int64_t n_groups, int pp, uint8_t* __restrict__ flags) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_groups && off[g + 1] - off[g] < pp) flags[g] = 1;
}
template <typename T>
__global__ void nan_flagged_kernel(const uint8_t* __restrict__ flags, int64_t n_groups, int pp, T* __restrict__ coeffs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_groups * pp && flags[i / pp]) coeffs[i] = (T)__builtin_nan("");
}

template <typename T>
static int grouped_big(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, const int64_t* d_off, int64_t n_groups, int64_t chunk,
                       const Method& method, const pds_lr_params* prm, const SolveParams& sp, T* d_mom, T* d_coeffs,
                       uint8_t* d_null) {
    const int bias = prm->add_bias ? 1 : 0, pp = n_feat + bias, q = n_feat + 2, nc = n_feat + 1;
    std::vector<int64_t> off((size_t)n_groups + 1);
    PDS_HIP_CHECK(hipMemcpyAsync(off.data(), d_off, off.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // one pointer table per group: the column bases moved to the group's first row
    std::vector<const T*> tables((size_t)n_groups * nc);
    for (int64_t g = 0; g < n_groups; ++g)
        for (int c = 0; c < nc; ++c) tables[(size_t)g * nc + c] = dc.h_ptrs[c] + off[g];
    const T** d_tables = reinterpret_cast<const T**>(ws_take(ctx, tables.size() * sizeof(T*)));
    PDS_HIP_CHECK(hipMemcpyAsync(d_tables, tables.data(), tables.size() * sizeof(T*), hipMemcpyHostToDevice, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const bool f32 = sizeof(T) == 4;
    for (int64_t g0 = 0; g0 < n_groups; g0 += chunk) {
        const int64_t gc = std::min(chunk, n_groups - g0);
        for (int64_t g = g0; g < g0 + gc; ++g) {
            T* rec = d_mom + (size_t)(g - g0) * q * q;
            const int64_t ng = off[g + 1] - off[g];
            if (ng <= 0) {
                PDS_HIP_CHECK(hipMemsetAsync(rec, 0, sizeof(T) * (size_t)q * q, ctx->stream));
                continue;
            }
            DeviceCols<T> dg;
            dg.nc = nc;
            dg.d_ptrs = d_tables + (size_t)g * nc;
            dg.h_ptrs.assign(tables.begin() + (size_t)g * nc, tables.begin() + (size_t)(g + 1) * nc);
            const size_t mark = ctx->ws_used;  // the SYRK partials are call-local: stream order makes the reuse safe
            const int rc = launch_moments_wide<T>(ctx, dg, n_feat, ng, false, rec);
            ctx->ws_used = mark;
            if (rc) return rc;
        }
        T* co = d_coeffs + g0 * pp;
        uint8_t* fl = d_null + g0;
        if (method.kind == Method::OLS) {
            if (int rc = launch_solve<T>(ctx, d_mom, gc, sp, co, fl, nullptr, nullptr)) return rc;
            // per-group pl_lr rejects "#Data < #features": null
            hipLaunchKernelGGL(mark_small_groups_kernel, dim3((unsigned)((gc + 255) / 256)), dim3(256), 0, ctx->stream, d_off + g0, gc,
                               pp, fl);
            hipLaunchKernelGGL((nan_flagged_kernel<T>), dim3((unsigned)((gc * pp + 255) / 256)), dim3(256), 0, ctx->stream, fl, gc, pp,
                               co);
            PDS_HIP_CHECK(hipGetLastError());
        } else if (method.kind == Method::NNLS) {
            if (int rc = launch_nnls<T>(ctx, d_mom, n_feat, bias, prm->tol, f32 ? 200 : prm->max_iter, co, gc, fl, d_off + g0)) return rc;
        } else if (int rc = launch_cd<T>(ctx, d_mom, n_feat, bias, method.l1, method.l2, prm->tol, f32 ? 2000 : prm->max_iter,
                                         method.positive, co, nullptr, gc, fl, d_off + g0))
            return rc;
    }
    return PDS_OK;
}

template <typename T>
static int grouped_impl(pds_ctx* ctx, const T* const* cols, int n_feat, int64_t n_rows, const int64_t* offsets,
                        int64_t n_groups, pds_space space, const pds_lr_params* prm, T* coeffs, uint8_t* is_null,
                        // nullable form: Arrow validity per column [y, x1..xp]; every group is fitted on the rows of it that
                        // survive the policy, like Polars calling pl_lr(null_policy=...) per group
                        bool nullable = false, const uint8_t* const* validity = nullptr, const int64_t* bit_offsets = nullptr,
                        int policy = PDS_NULL_RAISE, T fill_value = T(0),
                        // grouped pl_lr_pred (grouped_pred.hip): per-row outputs in the frame's row order, `space`-resident, each
                        // nullable; `coeffs` may then be null as well.  Not with the nullable form (its rows are compacted).
                        T* pred = nullptr, T* resid = nullptr, uint8_t* row_null = nullptr) {
    const bool want_pred = pred || resid || row_null;
    g_grouped_route = 0;  // (the hook reports THIS call's route: the routes that do not set it leave 0, not the previous call's value)
    if (!ctx || !cols || !offsets || !prm || (!coeffs && !want_pred)) return fail(PDS_ERR_INVALID, "null argument");
    if (want_pred && nullable) return fail(PDS_ERR_UNSUPPORTED, "grouped pred: null-free frames only");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_groups <= 0 || n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    const Method method = pick_method(prm);  // per group what pl_lr does per call: linear_regression.rs:447-497
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int bias = prm->add_bias ? 1 : 0, pp = n_feat + bias, q = n_feat + 2;
    // chunk the groups so one chunk's moment records (q*q values per group) stay inside the 256 MiB
    // Infinity Cache between the Gram kernel that writes them and the solve kernel that reads them
    const bool big = n_feat > kMaxFeatWide;  // > 64 features: one tiled-SYRK Gram build per group + solve_big
    // (17 .. 64 features: 2 GiB of records per chunk -- every chunk ends in a host synchronisation for the solver's marked count:
    //  100 000 groups x 100 rows x 64 features 7.4 ms in seven chunks of 512 MiB, 6.5 ms in two; PDS_GROUPED_CHUNK_MB overrides)
    static const int64_t chunk_env = [] { const char* e = dev_env("PDS_GROUPED_CHUNK_MB"); return e ? std::max<int64_t>(1, std::atoll(e)) : 0; }();
    const int64_t chunk_mb = chunk_env ? chunk_env : ((n_feat > 16 && n_feat <= kMaxFeatWide) ? 2048 : 128);
    int64_t chunk = std::max<int64_t>(big ? 64 : 4096, (int64_t)(chunk_mb << 20) / (int64_t)(sizeof(T) * q * q));
    chunk = std::min(chunk, n_groups);
    size_t need = 131072 + sizeof(T) * (size_t)chunk * q * q;
    need += (size_t)n_groups * 4 + (size_t)chunk * (pp * sizeof(T) + 1) + 4096;  // the fused path's pivoted-QR pass: list, results
    if (n_feat > 16 && n_feat <= 64) need += solve_wave_workspace(n_feat, bias, chunk, sizeof(T)) + 512;
    if (n_feat > 16 && n_feat <= 32) need += grouped_mid_fused_workspace(ctx->num_cus, n_feat, bias) + 512;
    if (big) need += (size_t)n_groups * (n_feat + 1) * sizeof(T*) + moments_wide_workspace(ctx->num_cus, n_feat, n_rows) + 8192;
    if (space == PDS_HOST || !coeffs) need += (size_t)(n_groups + 1) * 8 + (size_t)n_groups * (pp * sizeof(T) + 1) + 4096;
    if (want_pred && space == PDS_HOST) need += 2 * ((size_t)n_rows * sizeof(T) + 256) + (size_t)n_rows + 256;
    if (nullable) {
        if (policy < PDS_NULL_RAISE || policy > PDS_NULL_IGNORE) return fail(PDS_ERR_INVALID, "Invalid NullPolicy.");
        need += (1 << 20) + null_policy_workspace(n_feat + 1, n_rows, sizeof(T)) + (size_t)(n_groups + 1) * 8 + 4096;
        if (space == PDS_HOST) need += (size_t)(n_feat + 1) * ((size_t)n_rows / 8 + 4096);
    }
    if (int rc = ws_reserve(ctx, need)) return rc;
    DeviceCols<T> dc;
    if (int rc = make_device_cols<T>(ctx, cols, (const T*)nullptr, n_feat, n_rows, space, dc)) return rc;
    const int64_t* d_off = offsets;
    T* d_coeffs = coeffs;
    uint8_t* d_null = is_null;
    // small host batches (the plugin layer's coalesced per-group calls): offsets go up through pinned memory, coefficients
    // and null flags come back in ONE copy -- every pageable hipMemcpyAsync is 5-10 us of a ~60 us call
    const size_t co_bytes = ((size_t)n_groups * pp * sizeof(T) + 255) & ~(size_t)255;
    const bool small_out = space == PDS_HOST && coeffs && !want_pred &&
                           co_bytes + (size_t)n_groups + (size_t)(n_groups + 1) * 8 <= ((size_t)48 << 10);
    if (space == PDS_DEVICE && !coeffs) {  // (pred only: the coefficients live in the workspace)
        char* blk = reinterpret_cast<char*>(ws_take(ctx, co_bytes + (size_t)n_groups));
        d_coeffs = reinterpret_cast<T*>(blk);
        if (!d_null) d_null = reinterpret_cast<uint8_t*>(blk + co_bytes);
    }
    if (space == PDS_HOST) {
        int64_t* t = reinterpret_cast<int64_t*>(ws_take(ctx, (size_t)(n_groups + 1) * 8));
        if (small_out) {
            if (int rc = ensure_pinned(ctx, (size_t)128 << 10)) return rc;
            char* pin_off = static_cast<char*>(ctx->pinned) + ((size_t)64 << 10);
            std::memcpy(pin_off, offsets, (size_t)(n_groups + 1) * 8);
            PDS_HIP_CHECK(hipMemcpyAsync(t, pin_off, (size_t)(n_groups + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        } else {
            PDS_HIP_CHECK(hipMemcpyAsync(t, offsets, (size_t)(n_groups + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        }
        d_off = t;
        char* blk = reinterpret_cast<char*>(ws_take(ctx, co_bytes + (size_t)n_groups));
        d_coeffs = reinterpret_cast<T*>(blk);
        d_null = reinterpret_cast<uint8_t*>(blk + co_bytes);
    } else if (!d_null) {
        d_null = reinterpret_cast<uint8_t*>(ws_take(ctx, (size_t)n_groups));
    }
    if (nullable) {
        const int nc = n_feat + 1;
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
        if (prep.dropped) {
            int64_t* off2 = reinterpret_cast<int64_t*>(ws_take(ctx, (size_t)(n_groups + 1) * 8));
            if (int rc = remap_group_offsets(ctx, d_off, n_groups, prep.d_rank, n_rows, prep.n_kept, off2)) return rc;
            d_off = off2;
        }
        DeviceCols<T> dk;
        dk.nc = nc;
        dk.h_ptrs.resize(nc);
        for (int c = 0; c < n_feat; ++c) dk.h_ptrs[c] = prep.cols[c + 1];
        dk.h_ptrs[n_feat] = prep.cols[0];
        dk.h_ptrs.resize(std::max(nc, 18), dk.h_ptrs[0]);
        dk.d_ptrs = reinterpret_cast<const T**>(ws_take(ctx, sizeof(T*) * dk.h_ptrs.size()));
        PDS_HIP_CHECK(hipMemcpyAsync(dk.d_ptrs, dk.h_ptrs.data(), sizeof(T*) * dk.h_ptrs.size(), hipMemcpyHostToDevice, ctx->stream));
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        dc = dk;
        n_rows = prep.n_kept;
        if (n_rows == 0) return fail(PDS_ERR_EMPTY, "Empty data");
    }
    T* d_mom = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * (size_t)chunk * q * q));
    // solver = "svd" travels on: the streaming kernels answer the clear systems (any factorisation agrees on them to rounding) and
    // mark the ones next to the rank gate, which launch_solve_marked puts through the reference's SVD gate (lr_solvers.rs:358-366)
    SolveParams sp{n_feat, bias, prm->solver, prm->l2_reg, prm->singular_x_tol, 0};
    {
        const char* piv0 = dev_env("PDS_GROUPED_PIVOTED");
        if (piv0 && piv0[0] == '1' && sp.solver == PDS_SOLVER_CHOLESKEY) sp.solver = PDS_SOLVER_QR;
    }
    // Default (rank gate on): ONE streaming kernel, Gram + in-register Cholesky, no moment records in HBM.
    // Gate off (singular_x_tol = 0) needs the pivoted QR to reproduce the reference's answers on rank-deficient
    // groups; that solver is register hungry and runs faster as its own kernel behind the grouped Gram build.
    // PDS_GROUPED_UNFUSED=1 / PDS_GROUPED_PIVOTED=1 force the two-kernel pipeline / the pivoted QR (development).
    const char* unfused_env = dev_env("PDS_GROUPED_UNFUSED");
    const char* piv_env = dev_env("PDS_GROUPED_PIVOTED");
    const bool want_piv = !(sp.gate_tol > 0.0) || (piv_env && piv_env[0] == '1');
    if (big) {
        if (int rc = grouped_big<T>(ctx, dc, n_feat, d_off, n_groups, chunk, method, prm, sp, d_mom, d_coeffs, d_null)) return rc;
    } else if (method.kind != Method::OLS) {
        // lasso / elastic net / positive fits per group: grouped Gram build, then one wavefront per group runs the
        // reference's coordinate descent (faer_coordinate_descent / faer_nn_lr) on that group's moment record
        const bool f32 = sizeof(T) == 4;
        for (int64_t g0 = 0; g0 < n_groups; g0 += chunk) {
            const int64_t gc = std::min(chunk, n_groups - g0);
            if (int rc = launch_grouped_moments<T>(ctx, dc, n_feat, d_off + g0, gc, d_mom)) return rc;
            if (method.kind == Method::NNLS) {
                if (int rc = launch_nnls<T>(ctx, d_mom, n_feat, bias, prm->tol, f32 ? 200 : prm->max_iter, d_coeffs + g0 * pp, gc,
                                            d_null + g0, d_off + g0))
                    return rc;
            } else if (int rc = launch_cd<T>(ctx, d_mom, n_feat, bias, method.l1, method.l2, prm->tol,
                                             f32 ? 2000 : prm->max_iter, method.positive, d_coeffs + g0 * pp, nullptr, gc,
                                             d_null + g0, d_off + g0))
                return rc;
        }
    } else if (n_feat <= 16 && !want_piv && !(unfused_env && unfused_env[0] == '1') && n_groups < (1ll << 31)) {
        if (int rc = launch_grouped_fused<T>(ctx, dc, n_feat, n_rows, d_off, n_groups, sp, d_coeffs, d_null, d_mom, chunk)) return rc;
    } else {
        // 17 .. 32 f64 features, rank gate on (round 4): one stream, the solves in the streaming waves, no moment records
        // (grouped_mid.hip); PDS_GROUPED_MID_FUSED=0: the record pipeline below (A/B); it also takes over when more systems sit
        // next to the gate than the fused form's marked list holds
        bool mid_done = false;
        {
            const char* mf = dev_env("PDS_GROUPED_MID_FUSED");
            if (n_feat > 16 && n_feat <= 32 && !want_piv && !(mf && mf[0] == '0')) {
                void* d_fws = ws_take(ctx, grouped_mid_fused_workspace(ctx->num_cus, n_feat, bias));
                const int rcf = launch_grouped_mid_fused<T>(ctx, dc, n_feat, n_rows, d_off, n_groups, sp, d_coeffs, d_null, d_fws);
                if (rcf == PDS_OK) mid_done = true;
                else if (rcf != PDS_ERR_UNSUPPORTED) return rcf;
                g_grouped_route = mid_done ? 1 : 2;  // (development hook: pds_debug_last_grouped_route)
            }
        }
        void* d_wave_ws = (!mid_done && n_feat > 16 && n_feat <= 64 && !want_piv) ? ws_take(ctx, solve_wave_workspace(n_feat, bias, chunk, sizeof(T))) : nullptr;
        for (int64_t g0 = 0; g0 < n_groups && !mid_done; g0 += chunk) {
            const int64_t gc = std::min(chunk, n_groups - g0);
            bool streamed = false;
            if constexpr (std::is_same<T, double>::value) {
                // 17 .. 64 f64 features: the chunk's records from ONE stream over its rows (grouped_mid.hip); PDS_GROUPED_STREAM=0: A/B
                static const bool stream_off = [] { const char* e = dev_env("PDS_GROUPED_STREAM"); return e && e[0] == '0'; }();
                // (below ~28 features the padded 32-wide stream costs more than the one-wave-per-group kernel saves: 2.7 against 2.5 ms at 20)
                if (n_feat >= 28 && n_feat <= 64 && !stream_off) {
                    if (int rc = launch_grouped_moments_stream(ctx, dc, n_feat, n_rows, d_off + g0, gc, d_mom)) return rc;
                    streamed = true;
                }
            }
            if (!streamed)
                if (int rc = launch_grouped_moments<T>(ctx, dc, n_feat, d_off + g0, gc, d_mom)) return rc;
            // 17 .. 64 features, gate on: one wave per system in registers (solve_wave.hip), the pivoted QR only for what it marks
            int rcw = want_piv ? PDS_ERR_UNSUPPORTED
                               : launch_solve_wave<T>(ctx, d_mom, gc, sp, d_coeffs + g0 * pp, d_null + g0, d_off + g0, d_wave_ws);
            if (rcw == PDS_ERR_UNSUPPORTED) rcw = launch_solve<T>(ctx, d_mom, gc, sp, d_coeffs + g0 * pp, d_null + g0, nullptr, d_off + g0);
            if (rcw) return rcw;
        }
    }
    if (want_pred) {
        T* d_pred = pred;
        T* d_resid = resid;
        uint8_t* d_rn = row_null;
        if (space == PDS_HOST) {
            if (pred) d_pred = reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T)));
            if (resid) d_resid = reinterpret_cast<T*>(ws_take(ctx, (size_t)n_rows * sizeof(T)));
            if (row_null) d_rn = reinterpret_cast<uint8_t*>(ws_take(ctx, (size_t)n_rows));
        }
        if (int rc = launch_grouped_pred<T>(ctx, dc.d_ptrs, n_feat, bias, n_rows, d_off, n_groups, d_coeffs, d_null, nullptr, d_pred,
                                            d_resid, d_rn))
            return rc;
        if (space == PDS_HOST) {
            if (pred) PDS_HIP_CHECK(hipMemcpyAsync(pred, d_pred, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
            if (resid) PDS_HIP_CHECK(hipMemcpyAsync(resid, d_resid, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
            if (row_null) PDS_HIP_CHECK(hipMemcpyAsync(row_null, d_rn, (size_t)n_rows, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    if (small_out) {
        char* pin = static_cast<char*>(ctx->pinned);
        PDS_HIP_CHECK(hipMemcpyAsync(pin, d_coeffs, co_bytes + (size_t)n_groups, hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        std::memcpy(coeffs, pin, (size_t)n_groups * pp * sizeof(T));
        if (is_null) std::memcpy(is_null, pin + co_bytes, (size_t)n_groups);
        return PDS_OK;
    }
    if (space == PDS_HOST) {
        if (coeffs) PDS_HIP_CHECK(hipMemcpyAsync(coeffs, d_coeffs, (size_t)n_groups * pp * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        if (is_null) PDS_HIP_CHECK(hipMemcpyAsync(is_null, d_null, (size_t)n_groups, hipMemcpyDeviceToHost, ctx->stream));
    }
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// weighted groups: per group faer_weighted_lr (lr_solvers.rs:386-409) -- X' W X = (sqrt(W) X)' (sqrt(W) X), so the frame
// is scaled once on the device (the bias becomes an explicit sqrt(w) column) and takes the unweighted, ungated grouped path
// ---------------------------------------------------------------------------------------------
template <typename T>
static int grouped_weighted_impl(pds_ctx* ctx, const T* const* cols, const T* weights, int n_feat, int64_t n_rows,
                                 const int64_t* offsets, int64_t n_groups, pds_space space, const pds_lr_params* prm, T* coeffs,
                                 uint8_t* is_null,
                                 // grouped pl_lr_pred with weights: pred = x . beta on the UNSCALED rows (linear_regression.rs:782-785),
                                 // frame order (or through d_perm), `space`-resident, each nullable
                                 T* pred = nullptr, T* resid = nullptr, uint8_t* row_null = nullptr, const uint32_t* d_perm = nullptr) {
    const bool want_pred = pred || resid || row_null;
    if (!ctx || !cols || !weights || !offsets || !prm || (!coeffs && !want_pred)) return fail(PDS_ERR_INVALID, "null argument");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_groups <= 0 || n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int bias = prm->add_bias ? 1 : 0, pf = n_feat + bias, nc_in = n_feat + 1;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t col_bytes = up((size_t)n_rows * sizeof(T));
    size_t need = col_bytes * (pf + 1) + up(sizeof(T*) * (size_t)std::max(nc_in, 18)) + up((size_t)n_groups) + 4096;
    if (space == PDS_HOST || !coeffs) need += up((size_t)n_groups * pf * sizeof(T));
    if (space == PDS_HOST) need += col_bytes * (nc_in + 1) + up((size_t)(n_groups + 1) * 8);
    if (want_pred && space == PDS_HOST) need += 2 * col_bytes + up((size_t)n_rows);
    // (its own workspace: pds_lr_by_key_* calls this with its sorted frame living in ctx->keyed)
    if (int rc = ensure_ws(ctx, ctx->wkeyed, need)) return rc;
    char* w = static_cast<char*>(ctx->wkeyed.ptr);
    auto take = [&](size_t b) { char* r = w; w += up(b); return r; };
    std::vector<const T*> src(nc_in);
    const T* d_w = weights;
    const int64_t* d_off = offsets;
    T* d_co = coeffs;
    uint8_t* d_nu = is_null;
    if (space == PDS_HOST) {
        for (int c = 0; c < nc_in; ++c) {
            T* d = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
            PDS_HIP_CHECK(hipMemcpyAsync(d, cols[c], (size_t)n_rows * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
            src[c] = d;
        }
        T* dw = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
        PDS_HIP_CHECK(hipMemcpyAsync(dw, weights, (size_t)n_rows * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        d_w = dw;
        int64_t* doff = reinterpret_cast<int64_t*>(take((size_t)(n_groups + 1) * 8));
        PDS_HIP_CHECK(hipMemcpyAsync(doff, offsets, (size_t)(n_groups + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        d_off = doff;
    } else {
        for (int c = 0; c < nc_in; ++c) src[c] = cols[c];
    }
    if (space == PDS_HOST || !coeffs) d_co = reinterpret_cast<T*>(take((size_t)n_groups * pf * sizeof(T)));
    if (space == PDS_HOST || !d_nu) d_nu = reinterpret_cast<uint8_t*>(take((size_t)n_groups));
    // scaled frame in reference order [y, x1..xp, (sqrt w)]
    std::vector<const T*> scaled(pf + 1);
    for (int c = 0; c < nc_in; ++c) {
        T* d = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
        if (int rc = launch_scale_sqrt_w<T>(ctx, src[c], d_w, n_rows, d)) return rc;
        scaled[c] = d;
    }
    if (bias) {
        T* d = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
        if (int rc = launch_scale_sqrt_w<T>(ctx, (const T*)nullptr, d_w, n_rows, d)) return rc;
        scaled[nc_in] = d;
    }
    pds_lr_params p2 = *prm;  // faer_weighted_lr: plain solve with `solver`, no gate, no penalties
    p2.add_bias = 0;
    p2.l1_reg = 0.0;
    p2.l2_reg = 0.0;
    p2.positive = 0;
    p2.singular_x_tol = 0.0;
    if (int rc = grouped_impl<T>(ctx, scaled.data(), pf, n_rows, d_off, n_groups, PDS_DEVICE, &p2, d_co, d_nu)) return rc;
    if (want_pred) {
        // device pointer table of the unscaled frame in kernel order (x_0 .. x_{p-1}, y)
        std::vector<const T*> tbl((size_t)std::max(nc_in, 18), src[0]);
        for (int c = 0; c < n_feat; ++c) tbl[c] = src[c + 1];
        tbl[n_feat] = src[0];
        const T** d_tbl = reinterpret_cast<const T**>(take(sizeof(T*) * tbl.size()));
        PDS_HIP_CHECK(hipMemcpyAsync(d_tbl, tbl.data(), sizeof(T*) * tbl.size(), hipMemcpyHostToDevice, ctx->stream));
        T* d_pred = pred;
        T* d_resid = resid;
        uint8_t* d_rn = row_null;
        if (space == PDS_HOST) {
            if (pred) d_pred = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
            if (resid) d_resid = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
            if (row_null) d_rn = reinterpret_cast<uint8_t*>(take((size_t)n_rows));
        }
        if (int rc = launch_grouped_pred<T>(ctx, d_tbl, n_feat, bias, n_rows, d_off, n_groups, d_co, d_nu, d_perm, d_pred, d_resid, d_rn))
            return rc;
        if (space == PDS_HOST) {
            if (pred) PDS_HIP_CHECK(hipMemcpyAsync(pred, d_pred, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
            if (resid) PDS_HIP_CHECK(hipMemcpyAsync(resid, d_resid, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
            if (row_null) PDS_HIP_CHECK(hipMemcpyAsync(row_null, d_rn, (size_t)n_rows, hipMemcpyDeviceToHost, ctx->stream));
        }
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (tbl: source of the table copy)
    }
    if (space == PDS_HOST) {
        if (coeffs) PDS_HIP_CHECK(hipMemcpyAsync(coeffs, d_co, (size_t)n_groups * pf * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        if (is_null) PDS_HIP_CHECK(hipMemcpyAsync(is_null, d_nu, (size_t)n_groups, hipMemcpyDeviceToHost, ctx->stream));
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return PDS_OK;
}

// ---------------------------------------------------------------------------------------------
// grouped by an int64 key column in any row order (keyed.hip brings the frame into key order on the device)
// ---------------------------------------------------------------------------------------------
// keyed_partition.hip's moment table -> coefficients: chunks of groups expanded to (p+2)^2 records (a chunk stays inside the
// Infinity Cache between the expansion and the solve), then what grouped_impl does with grouped Gram records -- the batched
// pivoted QR / Cholesky with the rank gate, coordinate descent or NNLS, per pl_lr's dispatch (linear_regression.rs:447-497)
template <typename T>
static int solve_partition_table(pds_ctx* ctx, const KeyedPartitionState& st, int n_feat, int64_t n_groups, const int64_t* d_offsets,
                                 const pds_lr_params* prm, T* d_coeffs, uint8_t* d_null) {
    const Method method = pick_method(prm);
    const int bias = prm->add_bias ? 1 : 0, pp = n_feat + bias, q = n_feat + 2;
    int64_t chunk = std::max<int64_t>(4096, (int64_t)(128ll << 20) / (int64_t)(sizeof(T) * q * q));
    chunk = std::min(chunk, n_groups);
    if (int rc = ws_reserve(ctx, 131072 + sizeof(T) * (size_t)chunk * q * q)) return rc;
    T* d_mom = reinterpret_cast<T*>(ws_take(ctx, sizeof(T) * (size_t)chunk * q * q));
    SolveParams sp{n_feat, bias, prm->solver == PDS_SOLVER_SVD ? PDS_SOLVER_QR : prm->solver, prm->l2_reg, prm->singular_x_tol, 0};
    const bool f32 = sizeof(T) == 4;
    // OLS / ridge with up to 16 coefficients: the register-resident solver reads the table's triangles itself (no expansion pass)
    if (method.kind == Method::OLS && pp <= 16 && (sp.solver != PDS_SOLVER_CHOLESKEY || sp.gate_tol > 0.0)) {
        static const bool expand = [] { const char* e = dev_env("PDS_PART_EXPAND"); return e && e[0] == '1'; }();  // (A/B)
        if (!expand) {
            TriSource tri;
            tri.table = st.table;
            tri.ids = st.ids;
            tri.nvp = st.nvp;
            tri.pc = st.pc;
            return launch_solve_reg<T>(ctx, nullptr, n_groups, sp, d_coeffs, d_null, d_offsets, &tri);
        }
    }
    for (int64_t g0 = 0; g0 < n_groups; g0 += chunk) {
        const int64_t gc = std::min(chunk, n_groups - g0);
        if (int rc = keyed_partition_records<T>(ctx, st, n_feat, g0, gc, d_mom)) return rc;
        KernelTimer timer(ctx, kKindSolve);
        if (method.kind == Method::OLS) {
            if (int rc = launch_solve<T>(ctx, d_mom, gc, sp, d_coeffs + g0 * pp, d_null + g0, nullptr, d_offsets + g0)) return rc;
        } else if (method.kind == Method::NNLS) {
            if (int rc = launch_nnls<T>(ctx, d_mom, n_feat, bias, prm->tol, f32 ? 200 : prm->max_iter, d_coeffs + g0 * pp, gc, d_null + g0,
                                        d_offsets + g0))
                return rc;
        } else if (int rc = launch_cd<T>(ctx, d_mom, n_feat, bias, method.l1, method.l2, prm->tol, f32 ? 2000 : prm->max_iter, method.positive,
                                         d_coeffs + g0 * pp, nullptr, gc, d_null + g0, d_offsets + g0)) {
            return rc;
        }
    }
    return PDS_OK;
}

// Where a slice of a frame that is fitted in several pieces (capi_multi.hpp) puts its results: asked once the slice's number of
// groups is known, before the fit.  `unsorted()` reports that the slice's keys are not in order (the pieces are then meaningless).
template <typename T>
struct ByKeyPlace {
    virtual int at(int64_t n_groups, int64_t** out_keys, T** coeffs, uint8_t** is_null) = 0;
    virtual void unsorted() = 0;
    virtual ~ByKeyPlace() = default;
};

template <typename T>
static int lr_by_key_impl(pds_ctx* ctx, const T* const* cols, const int64_t* keys, int n_feat, int64_t n_rows, pds_space space,
                          const pds_lr_params* prm, int64_t max_groups, int64_t* out_keys, T* coeffs, uint8_t* is_null,
                          int64_t* n_groups,
                          // pds_lr_by_key_pred_*: optional weights (one more column through the ordering), per-row outputs in the
                          // FRAME's row order; out_keys / coeffs / is_null / n_groups are then optional
                          const T* weights = nullptr, T* pred = nullptr, T* resid = nullptr, uint8_t* row_null = nullptr,
                          ByKeyPlace<T>* place = nullptr) {
    const bool want_pred = pred || resid || row_null;
    const bool want_coef = out_keys || coeffs || place;
    if (!ctx || !cols || !keys || !prm) return fail(PDS_ERR_INVALID, "null argument");
    if (!want_pred && !place && (!out_keys || !coeffs || !n_groups)) return fail(PDS_ERR_INVALID, "null argument");
    if (place && space != PDS_HOST) return fail(PDS_ERR_INVALID, "sliced fits take host frames");
    if (want_coef && !place && (!out_keys || !coeffs)) return fail(PDS_ERR_INVALID, "out_keys and coeffs come together");
    if (n_feat < 1) return fail(PDS_ERR_INVALID, "need at least one feature column");
    if (n_rows <= 0) return fail(PDS_ERR_EMPTY, "Empty data");
    if (!want_coef) max_groups = n_rows;
    if (max_groups < 1) return fail(PDS_ERR_INVALID, "max_groups must be positive");
    PDS_HIP_CHECK(hipSetDevice(ctx->device));
    const int nc = n_feat + 1 + (weights ? 1 : 0), pp = n_feat + (prm->add_bias ? 1 : 0);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t key_bytes = up((size_t)n_rows * 8), col_bytes = up((size_t)n_rows * sizeof(T)), idx_bytes = up((size_t)n_rows * 4);
    StageTrace tr(ctx, "pds_lr_by_key");
    // ---- keys on the device, and are they already in order?
    const int64_t* d_keys = keys;
    if (space == PDS_HOST) {
        if (int rc = ensure_ws(ctx, ctx->stage, key_bytes + 256)) return rc;
        PDS_HIP_CHECK(hipMemcpyAsync(ctx->stage.ptr, keys, (size_t)n_rows * 8, hipMemcpyHostToDevice, ctx->stream));
        d_keys = static_cast<const int64_t*>(ctx->stage.ptr);
    }
    if (int rc = ensure_pinned(ctx, 4096)) return rc;
    // order flag + key range + the run counts of the order check (keyed.hip): they outlive the workspace sizing below
    const size_t run_slots = key_run_slots(n_rows);
    const size_t slot_bytes = (size_t)kKeySlots * 8 * sizeof(unsigned);
    const size_t mask_bytes = key_run_mask_bytes(n_rows);
    if (int rc = ensure_ws(ctx, ctx->solve_ws, 8192 + 2 * up((run_slots + 1) * sizeof(uint32_t)) + up(slot_bytes) + mask_bytes)) return rc;
    bool sorted = false;
    int64_t mm[2] = {0, 0};
    int64_t n_runs = 0;  // keys that differ from their successor: n_groups - 1 of an ordered column
    int64_t* d_state = reinterpret_cast<int64_t*>(static_cast<char*>(ctx->solve_ws.ptr) + 256);
    int64_t* d_minmax = d_state + 2;
    uint32_t* d_run_counts = reinterpret_cast<uint32_t*>(static_cast<char*>(ctx->solve_ws.ptr) + 4096);
    uint32_t* d_run_prefix = reinterpret_cast<uint32_t*>(static_cast<char*>(ctx->solve_ws.ptr) + 4096 + up((run_slots + 1) * sizeof(uint32_t)));
    unsigned long long* d_run_masks = reinterpret_cast<unsigned long long*>(static_cast<char*>(ctx->solve_ws.ptr) + 4096 +
                                                                            2 * up((run_slots + 1) * sizeof(uint32_t)) + up(slot_bytes));
    // dense-key candidates (unweighted, <= 16 features): the order check takes the partition route's bucket histogram along
    // context option "keyed_sort" (default from PDS_KEYED_SORT=1 at pds_ctx_create): the sorting route for every unordered frame -- the DETERMINISM switch: the partition route's
    // record order follows cursor atomics, so its sums are reproducible to rounding only.  solver = "svd" with the
    // rank gate on also sorts: the partition route solves every group with the pivoted QR and marks nothing for the SVD gate.
    const bool force_sort = ctx->opt_keyed_sort || (prm->solver == PDS_SOLVER_SVD && prm->singular_x_tol > 0.0);
    const bool part_candidate = !force_sort && !weights && n_feat <= 16 && !place;
    const int part_shift = part_candidate ? keyed_partition_shift<T>(n_feat) : -1;
    unsigned* d_slots = reinterpret_cast<unsigned*>(static_cast<char*>(ctx->solve_ws.ptr) + 4096 + 2 * up((run_slots + 1) * sizeof(uint32_t)));
    bool hist_taken = false;
    if (int rc = keys_order_minmax(ctx, d_keys, n_rows, d_state, &sorted, mm, d_run_counts, d_run_masks, &n_runs, part_shift,
                                   part_candidate ? d_slots : nullptr, &hist_taken))
        return rc;
    tr.mark("keys H2D + order check");
    // ORDERED keys have no row bound of their own (the order check, the run marks and the fits index rows with 64 bits; 2^31 + rows x 8
    // f64 features fit this device's HBM, and the reference's series_to_mat_for_lr has no bound either, linear_regression.rs:151-267);
    // the routes for keys in ANY order carry 32-bit row ranks through the sort / the partition
    if (!sorted && n_rows >= (1ll << 31)) return fail(PDS_ERR_UNSUPPORTED, "keyed grouping of unordered keys: fewer than 2^31 rows per call");
    if (place && !sorted) {
        place->unsorted();
        return fail(PDS_ERR_UNSUPPORTED, "sliced fit: the slice's keys are not in order");
    }
    // ---- keys in any order: smallest / largest key decide the route.  Dense integer keys (group ids) of an unweighted
    // fit with up to 16 features take the partition route (keyed_partition.hip: no sort, no random-access pass; per-row
    // predictions then look the row's group up from its key -- grouped_pred.hip MODE 2 -- and nothing is ever permuted);
    // PDS_KEYED_SORT=1 keeps the sorting route (A/B)
    int64_t part_buckets = 0;
    int64_t part_base = 0;       // smallest key rounded down to a multiple of the bucket width: the origin of the dense ids
    uint64_t part_range = 0;
    if (!sorted && part_candidate) {
        const int64_t wdt = (int64_t)1 << part_shift;
        part_base = mm[0] - (((mm[0] % wdt) + wdt) % wdt);
        part_range = (uint64_t)mm[1] - (uint64_t)part_base + 1;
        part_buckets = keyed_partition_buckets<T>(n_feat, n_rows, part_range);
    }
    const bool partition = part_buckets > 0;
    int64_t* d_part_base = d_state + 1;  // (the unused slot of the order check's state)
    if (partition) PDS_HIP_CHECK(hipMemcpyAsync(d_part_base, &part_base, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    // ---- workspace: [raw columns (host frames)] [sorted keys, index in/out, gathered columns (unsorted frames)] runs, temp
    const int64_t cap = std::min<int64_t>(max_groups, n_rows);
    const size_t temp_bytes = sorted ? keyed_ordered_temp_bytes(n_rows) : keyed_temp_bytes(n_rows);
    // unique keys, counts, offsets: at most one per group (the partition route; ordered keys: the order check has counted them) / per row
    const int64_t run_cap = partition ? cap : (sorted ? std::min<int64_t>(n_runs + 1, cap) : n_rows);
    size_t need = temp_bytes + 3 * up((size_t)(run_cap + 1) * 8) + 8192;
    if (space == PDS_HOST) need += col_bytes * nc;
    // (the same predicates as the take() sites below: a device frame with coeffs but no is_null still takes its flags here)
    if (space == PDS_HOST || !coeffs) need += up((size_t)cap * pp * sizeof(T));
    if (space == PDS_HOST || !is_null) need += up((size_t)cap);
    if (partition) need += keyed_partition_workspace<T>(n_feat, n_rows, part_buckets) + up(sizeof(T*) * (size_t)std::max(nc, 18));
    // per-row predictions on the partition route: an id-indexed copy of the coefficient block (one randomly read object per row) when
    // the id space is small enough to live in the memory-side cache like the block itself -- up to 2^25 ids and 1 GiB
    const int64_t part_ids = partition ? keyed_partition_table_ids<T>(n_feat, part_buckets) : 0;
    const size_t cbi_bytes = (size_t)part_ids * grouped_pred_table_stride<T>(pp) * sizeof(T);
    const bool pred_table = partition && want_pred && part_ids <= ((int64_t)1 << 25) && cbi_bytes <= ((size_t)1 << 30);
    if (pred_table) need += up(cbi_bytes);
    // the sort + gather route's buffers: only when neither the order check nor the partition route serves the frame
    if (!partition && !sorted) need += 2 * key_bytes + 2 * idx_bytes + col_bytes * nc + up((size_t)n_rows * nc * sizeof(T)) + up(2 * (size_t)nc * sizeof(T*)) + 1024;
    if (want_pred) need += up(sizeof(T*) * (size_t)std::max(nc, 18)) + (space == PDS_HOST ? 2 * col_bytes + up((size_t)n_rows) : 0);
    if (int rc = ensure_ws(ctx, ctx->keyed, need)) return rc;
    tr.mark("workspace");
    char* w = static_cast<char*>(ctx->keyed.ptr);
    auto take = [&](size_t b) { char* r = w; w += up(b); return r; };
    void* d_temp = take(temp_bytes);
    int64_t* d_unique = reinterpret_cast<int64_t*>(take((size_t)(run_cap + 1) * 8));
    int64_t* d_counts = reinterpret_cast<int64_t*>(take((size_t)(run_cap + 1) * 8));
    int64_t* d_offsets = reinterpret_cast<int64_t*>(take((size_t)(run_cap + 1) * 8));
    int64_t* d_nruns = reinterpret_cast<int64_t*>(take(256));
    std::vector<const T*> src(nc);  // reference order [y, x1..xp, (w)], device resident
    for (int c = 0; c < n_feat + 1; ++c) src[c] = cols[c];
    if (weights) src[n_feat + 1] = weights;
    if (space == PDS_HOST)
        for (int c = 0; c < nc; ++c) {
            T* d = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
            PDS_HIP_CHECK(hipMemcpyAsync(d, src[c], (size_t)n_rows * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
            src[c] = d;
        }
    tr.mark("columns H2D");
    if (partition) {
        std::vector<const T*> tbl((size_t)std::max(nc, 18), src[0]);
        for (int c = 0; c < n_feat; ++c) tbl[c] = src[c + 1];
        tbl[n_feat] = src[0];
        const T** d_tbl = reinterpret_cast<const T**>(take(sizeof(T*) * tbl.size()));
        PDS_HIP_CHECK(hipMemcpyAsync(d_tbl, tbl.data(), sizeof(T*) * tbl.size(), hipMemcpyHostToDevice, ctx->stream));
        char* pws = take(keyed_partition_workspace<T>(n_feat, n_rows, part_buckets));
        KeyedPartitionState st;
        int64_t ng = 0;
        const bool use_slots = hist_taken && part_buckets <= kKeySlots;
        const unsigned first_slot = (unsigned)((part_base >> part_shift) & (int64_t)(kKeySlots - 1));
        const int rc0 = keyed_partition_build<T>(ctx, d_tbl, d_keys, d_part_base, part_range, n_feat, n_rows, part_buckets, pws, cap, d_unique,
                                                 d_offsets, &ng, st, use_slots ? d_slots : nullptr, first_slot);
        if (n_groups) *n_groups = ng;
        if (rc0) return rc0;
        tr.mark("partition + moments");
        if (place)
            if (int rc = place->at(ng, &out_keys, &coeffs, &is_null)) return rc;
        T* d_co = coeffs;
        uint8_t* d_nu = is_null;
        if (space == PDS_HOST || !coeffs) d_co = reinterpret_cast<T*>(take((size_t)cap * pp * sizeof(T)));
        if (space == PDS_HOST || !is_null) d_nu = reinterpret_cast<uint8_t*>(take((size_t)cap));
        if (int rc = solve_partition_table<T>(ctx, st, n_feat, ng, d_offsets, prm, d_co, d_nu)) return rc;
        tr.mark("solve");
        if (want_pred) {
            T* d_pred = pred;
            T* d_resid = resid;
            uint8_t* d_rn = row_null;
            if (space == PDS_HOST) {
                if (pred) d_pred = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
                if (resid) d_resid = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
                if (row_null) d_rn = reinterpret_cast<uint8_t*>(take((size_t)n_rows));
            }
            if (pred_table) {
                T* d_cbi = reinterpret_cast<T*>(take(cbi_bytes));
                if (int rc = launch_grouped_pred_by_id_table<T>(ctx, d_tbl, n_feat, prm->add_bias ? 1 : 0, n_rows, d_keys, d_part_base, st.ids, ng, d_co,
                                                                d_nu, d_cbi, d_pred, d_resid, d_rn))
                    return rc;
            } else if (int rc = launch_grouped_pred_by_id<T>(ctx, d_tbl, n_feat, prm->add_bias ? 1 : 0, n_rows, d_keys, d_part_base, st.rank, ng, d_co,
                                                             d_nu, d_pred, d_resid, d_rn)) {
                return rc;
            }
            if (space == PDS_HOST) {
                if (pred) PDS_HIP_CHECK(hipMemcpyAsync(pred, d_pred, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
                if (resid) PDS_HIP_CHECK(hipMemcpyAsync(resid, d_resid, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
                if (row_null) PDS_HIP_CHECK(hipMemcpyAsync(row_null, d_rn, (size_t)n_rows, hipMemcpyDeviceToHost, ctx->stream));
            }
            tr.mark("pred");
        }
        if (space == PDS_HOST) {
            if (coeffs) PDS_HIP_CHECK(hipMemcpyAsync(coeffs, d_co, (size_t)ng * pp * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
            if (coeffs && is_null) PDS_HIP_CHECK(hipMemcpyAsync(is_null, d_nu, (size_t)ng, hipMemcpyDeviceToHost, ctx->stream));
            if (out_keys) PDS_HIP_CHECK(hipMemcpyAsync(out_keys, d_unique, (size_t)ng * 8, hipMemcpyDeviceToHost, ctx->stream));
        } else if (out_keys) {
            PDS_HIP_CHECK(hipMemcpyAsync(out_keys, d_unique, (size_t)ng * 8, hipMemcpyDeviceToDevice, ctx->stream));
        }
        PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        tr.mark("results D2H");
        return PDS_OK;
    }
    const int64_t* d_sorted_keys = d_keys;
    const uint32_t* d_perm = nullptr;
    if (!sorted) {
        int64_t* sk = reinterpret_cast<int64_t*>(take((size_t)n_rows * 8));
        uint32_t* idx_in = reinterpret_cast<uint32_t*>(take((size_t)n_rows * 4));
        uint32_t* perm = reinterpret_cast<uint32_t*>(take((size_t)n_rows * 4));
        int64_t* sk2 = reinterpret_cast<int64_t*>(take((size_t)n_rows * 8));
        if (int rc = keyed_sort(ctx, d_keys, n_rows, idx_in, sk, perm, d_temp, temp_bytes, sk2, d_minmax, mm)) return rc;
        d_sorted_keys = sk;
        d_perm = perm;
        static const bool by_column = [] { const char* e = dev_env("PDS_KEYED_GATHER_BY_COLUMN"); return e && e[0] == '1'; }();
        // frames too wide for the 256-row transposition tile (32 f64 / 64 f32 columns and beyond) gather column by column
        if (by_column || !gather_frame_fits<T>(nc)) {  // (one random 8-byte read per element; the env switch is the A/B)
            for (int c = 0; c < nc; ++c) {
                T* d = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
                if (int rc = launch_gather_rows<T>(ctx, src[c], perm, n_rows, d)) return rc;
                src[c] = d;
            }
        } else {
            // transpose to row-major records, then one random access per ROW (keyed.hip)
            std::vector<const T*> tbl(2 * (size_t)nc);
            for (int c = 0; c < nc; ++c) tbl[c] = src[c];
            for (int c = 0; c < nc; ++c) {
                T* d = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
                tbl[nc + c] = d;
                src[c] = d;
            }
            T* records = reinterpret_cast<T*>(take((size_t)n_rows * nc * sizeof(T)));
            const T** d_tbl = reinterpret_cast<const T**>(take(2 * (size_t)nc * sizeof(T*)));
            PDS_HIP_CHECK(hipMemcpyAsync(d_tbl, tbl.data(), 2 * (size_t)nc * sizeof(T*), hipMemcpyHostToDevice, ctx->stream));
            if (int rc = launch_gather_frame<T>(ctx, d_tbl, perm, nc, n_rows, records, (T* const*)(d_tbl + nc)))
                return rc;
            PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (tbl: source of the table copy)
        }
    }
    tr.mark("sort + gather");
    int64_t ng = 0;
    if (sorted) {  // keys in order: the order check has counted and marked the run starts already -- a scan and one pass over the marks,
        ng = n_runs + 1;  // left on the stream in front of the fit (the number of groups came back with the order flag)
        if (ng <= max_groups)
            if (int rc = keyed_runs_ordered(ctx, d_keys, n_rows, d_run_counts, d_run_prefix, d_run_masks, run_cap, d_unique, d_offsets, d_temp, temp_bytes))
                return rc;
    } else if (int rc = keyed_runs(ctx, d_sorted_keys, n_rows, d_unique, d_counts, d_offsets, d_nruns, d_temp, temp_bytes, &ng)) {
        return rc;
    }
    tr.mark("run lengths + offsets");
    if (n_groups) *n_groups = ng;
    if (ng > max_groups) return fail(PDS_ERR_INVALID, "more distinct keys than max_groups");
    if (place)
        if (int rc = place->at(ng, &out_keys, &coeffs, &is_null)) return rc;
    T* d_co = coeffs;
    uint8_t* d_nu = is_null;
    if (space == PDS_HOST || !coeffs) d_co = reinterpret_cast<T*>(take((size_t)cap * pp * sizeof(T)));
    if (space == PDS_HOST || !is_null) d_nu = reinterpret_cast<uint8_t*>(take((size_t)cap));
    T* d_pred = pred;
    T* d_resid = resid;
    uint8_t* d_rn = row_null;
    if (want_pred && space == PDS_HOST) {
        if (pred) d_pred = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
        if (resid) d_resid = reinterpret_cast<T*>(take((size_t)n_rows * sizeof(T)));
        if (row_null) d_rn = reinterpret_cast<uint8_t*>(take((size_t)n_rows));
    }
    if (weights) {
        if (int rc = grouped_weighted_impl<T>(ctx, src.data(), src[n_feat + 1], n_feat, n_rows, d_offsets, ng, PDS_DEVICE, prm, d_co, d_nu,
                                              d_pred, d_resid, d_rn, d_perm))
            return rc;
    } else {
        if (int rc = grouped_impl<T>(ctx, src.data(), n_feat, n_rows, d_offsets, ng, PDS_DEVICE, prm, d_co, d_nu)) return rc;
        if (want_pred) {
            std::vector<const T*> tbl((size_t)std::max(nc, 18), src[0]);
            for (int c = 0; c < n_feat; ++c) tbl[c] = src[c + 1];
            tbl[n_feat] = src[0];
            const T** d_tbl = reinterpret_cast<const T**>(take(sizeof(T*) * tbl.size()));
            PDS_HIP_CHECK(hipMemcpyAsync(d_tbl, tbl.data(), sizeof(T*) * tbl.size(), hipMemcpyHostToDevice, ctx->stream));
            if (int rc = launch_grouped_pred<T>(ctx, d_tbl, n_feat, prm->add_bias ? 1 : 0, n_rows, d_offsets, ng, d_co, d_nu, d_perm, d_pred,
                                                d_resid, d_rn))
                return rc;
            PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (tbl: source of the table copy)
        }
    }
    tr.mark("grouped fit");
    if (space == PDS_HOST) {
        if (coeffs) PDS_HIP_CHECK(hipMemcpyAsync(coeffs, d_co, (size_t)ng * pp * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        if (coeffs && is_null) PDS_HIP_CHECK(hipMemcpyAsync(is_null, d_nu, (size_t)ng, hipMemcpyDeviceToHost, ctx->stream));
        if (out_keys) PDS_HIP_CHECK(hipMemcpyAsync(out_keys, d_unique, (size_t)ng * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (pred) PDS_HIP_CHECK(hipMemcpyAsync(pred, d_pred, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        if (resid) PDS_HIP_CHECK(hipMemcpyAsync(resid, d_resid, (size_t)n_rows * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        if (row_null) PDS_HIP_CHECK(hipMemcpyAsync(row_null, d_rn, (size_t)n_rows, hipMemcpyDeviceToHost, ctx->stream));
    } else if (out_keys) {
        PDS_HIP_CHECK(hipMemcpyAsync(out_keys, d_unique, (size_t)ng * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    tr.mark("results D2H");
    return PDS_OK;
}
