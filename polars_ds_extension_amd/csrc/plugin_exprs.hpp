// plugin_exprs.hpp -- the expression bodies: pl_lr(_pred), pl_lin_reg_report / pl_wls_report, pl_rolling_lr / pl_recursive_lr, pl_lr_by, pl_lr_multi(_pred), pl_lr_w_rcond
// Part of the one translation unit plugin.cpp (included there, inside its anonymous namespace, in dependency order).
#pragma once

// ------------------------------------------------------------------------------------------------- pl_lr / pl_lr_pred
template <typename T>
void do_pl_lr(SeriesExport* in, size_t n_in, const Kwargs& kw, SeriesExport* out, bool want_pred) {
    const pds_lr_params prm = lr_params(kw);
    const Policy pol = parse_policy(kw_str(kw, "null_policy", "raise"));
    const bool weighted = kw_bool(kw, "weighted");
    auto cols = import_all<T>(in, n_in);
    const size_t first = weighted ? 1 : 0;
    if (n_in < first + 2) raise("need a target and at least one feature");
    const int n_feat = (int)(n_in - first - 1);
    const int64_t n = cols[first].size();
    for (auto& c : cols)
        if (c.size() != n) raise("input columns differ in length");
    const int pp = n_feat + prm.add_bias;
    std::vector<T> coeffs(pp);
    int is_null = 0;
    ByteVec pred_b, resid_b;  // result storage: written by the copy back from the device, handed to the Arrow arrays
    std::vector<uint8_t> valid;
    bool any_null = false;
    for (auto& c : cols) any_null |= c.null_count > 0;
    if (want_pred && !weighted && any_null && pol.kind == Policy::IGNORE && reference_quirks()) {
        // the reference's answer (see reference_quirks()): one all-null row, whatever the frame holds
        const uint8_t none = 0;
        const T zero = 0;
        std::vector<std::unique_ptr<ArrowArray>> kids;
        kids.push_back(prim_array<T>(&zero, 1, &none));
        kids.push_back(prim_array<T>(&zero, 1, &none));
        std::vector<std::unique_ptr<ArrowSchema>> sk;
        sk.push_back(make_schema(fmt_of<T>(), "pred"));
        sk.push_back(make_schema(fmt_of<T>(), "resid"));
        export_series(out, make_schema("+s", "", std::move(sk)), struct_array(1, std::move(kids)));
        return;
    }
    // small frames (the per-group calls of group_by().agg() / .over()): leave together with whatever else is queued.  Eligible:
    // coefficient and pred fits without nulls (weighted or not), coefficient fits with nulls under skip / fill whose kept rows
    // still cover the coefficients (fewer kept rows than coefficients is an ERROR of the single-frame path, not a null: it keeps it)
    int queue_null_code = 0;
    bool queue_ok = n >= pp && n > 0 && (size_t)n * (n_feat + 1 + first) * sizeof(T) <= ((size_t)256 << 10) && n_feat <= 64 &&
                    prm.solver != PDS_SOLVER_SVD && coalescing_enabled();
    if (queue_ok && any_null) {
        queue_ok = !weighted && !want_pred && (pol.kind == Policy::SKIP || pol.kind == Policy::FILL);
        if (queue_ok) {
            int64_t kept = 0;
            for (int64_t i = 0; i < n; ++i) {
                bool ok = true;
                const size_t last = pol.kind == Policy::SKIP ? n_in : first + 1;  // (fill: only the target's nulls drop rows)
                for (size_t c = first; c < last && ok; ++c) ok = cols[c].validity.empty() || bit_get(cols[c].validity.data(), i);
                kept += ok ? 1 : 0;
            }
            queue_ok = kept >= pp;
            queue_null_code = pol.kind == Policy::SKIP ? PDS_NULL_SKIP : PDS_NULL_FILL;
        }
    }
    if (queue_ok) {
        LrRequest<T> req;
        req.cols = &cols;
        req.first = first;
        req.n_feat = n_feat;
        req.n = n;
        req.prm = prm;
        req.weighted = weighted;
        req.want_pred = want_pred;
        req.null_code = queue_null_code;
        req.fill = (T)pol.fill;
        req.coeffs = &coeffs;
        if (want_pred) {
            pred_b = raw_buffer<T>((size_t)n);
            resid_b = raw_buffer<T>((size_t)n);
            req.pred = as<T>(pred_b);
            req.resid = as<T>(resid_b);
            valid.assign(n, 1);
        }
        if (settings().coalesce == 2) LrCoalescer<T>::run_one(&req);
        else LrCoalescer<T>::instance().submit(&req);
        if (!req.error.empty()) raise(req.error);
        is_null = req.is_null;
    } else if (weighted) {
        // weights are input 0 and go through the null-free entry point (faer_weighted_lr ignores l1/l2 and the gate)
        for (auto& c : cols)
            if (c.null_count) raise("Nulls found in data");
        std::vector<const T*> ptrs;
        for (size_t i = first; i < n_in; ++i) ptrs.push_back(cols[i].data());
        if (want_pred) {
            pred_b = raw_buffer<T>((size_t)n);
            resid_b = raw_buffer<T>((size_t)n);
            check(Api<T>::lr_pred(thread_ctx(), ptrs.data(), cols[0].data(), n_feat, n, PDS_HOST, &prm, coeffs.data(),
                                  &is_null, as<T>(pred_b), as<T>(resid_b)));
            valid.assign(n, 1);
        } else {
            check(Api<T>::lr(thread_ctx(), ptrs.data(), cols[0].data(), n_feat, n, PDS_HOST, &prm, coeffs.data(), &is_null));
        }
    } else {
        std::vector<const T*> ptrs;
        std::vector<const uint8_t*> bms;
        std::vector<int64_t> offs;
        for (size_t i = first; i < n_in; ++i) {
            ptrs.push_back(cols[i].data());
            bms.push_back(cols[i].validity.empty() ? nullptr : cols[i].validity.data());
            offs.push_back(0);
        }
        int code = pol.kind == Policy::RAISE ? PDS_NULL_RAISE : pol.kind == Policy::FILL ? PDS_NULL_FILL
                   : pol.kind == Policy::SKIP ? PDS_NULL_SKIP : PDS_NULL_IGNORE;
        int64_t n_used = 0;
        if (want_pred) {
            pred_b = raw_buffer<T>((size_t)n);
            resid_b = raw_buffer<T>((size_t)n);
            valid.resize(n);
        }
        check(Api<T>::lr_nullable(thread_ctx(), ptrs.data(), bms.data(), offs.data(), n_feat, n, PDS_HOST, code, (T)pol.fill, &prm,
                                  coeffs.data(), &is_null, want_pred ? as<T>(pred_b) : nullptr, want_pred ? as<T>(resid_b) : nullptr,
                                  want_pred ? valid.data() : nullptr, &n_used));
    }
    if (!want_pred) {
        // 1-row List<T> "coeffs"; a gated fit is a 1-row null list (:462-468, :356-365)
        const uint8_t ok = is_null ? 0 : 1;
        std::vector<int64_t> off = {0, is_null ? 0 : (int64_t)pp};
        export_series(out, list_schema<T>("coeffs"), list_array<T>(off, &ok, coeffs.data(), is_null ? 0 : pp));
        return;
    }
    if (is_null) std::fill(valid.begin(), valid.end(), (uint8_t)0);  // all-null {pred, resid} (:745-750)
    std::vector<std::unique_ptr<ArrowArray>> kids;
    kids.push_back(prim_array_take<T>(std::move(pred_b), n, valid.data()));
    kids.push_back(prim_array_take<T>(std::move(resid_b), n, valid.data()));
    std::vector<std::unique_ptr<ArrowSchema>> sk;
    sk.push_back(make_schema(fmt_of<T>(), "pred"));
    sk.push_back(make_schema(fmt_of<T>(), "resid"));
    export_series(out, make_schema("+s", "", std::move(sk)), struct_array(n, std::move(kids)));
}

// ------------------------------------------------------------------------------------------------- lin_reg_report
template <typename T>
void do_report(SeriesExport* in, size_t n_in, const Kwargs& kw, SeriesExport* out, bool wls) {
    // inputs: [weights?, var(y), y, x1..xp]  (expr_linear.py:614-617)
    auto cols = import_all<T>(in, n_in);
    const size_t iy = wls ? 2 : 1;
    if (n_in < iy + 2) raise("need a target and at least one feature");
    const bool bias = kw_bool(kw, "bias");
    const std::string se = kw_str(kw, "std_err", "se");
    const int se_type = se == "hc0" ? PDS_HC0 : se == "hc1" ? PDS_HC1 : se == "hc2" ? PDS_HC2 : se == "hc3" ? PDS_HC3 : PDS_SE;
    const Policy pol = parse_policy(kw_str(kw, "null_policy", "raise"));
    bool any_null = false;
    for (size_t i = iy; i < n_in; ++i) any_null |= cols[i].null_count > 0;
    // pl_wls_report never compacts its weights (linear_regression.rs:1013-1020): nulls there cannot work in the reference either
    if (any_null && (wls || pol.kind == Policy::RAISE)) raise("Nulls found in data");
    const T y_var = cols[iy - 1].size() == 0 || cols[iy - 1].null_count ? (T)NAN : cols[iy - 1].at(0);
    const int n_feat = (int)(n_in - iy - 1);
    const int pp = n_feat + (bias ? 1 : 0);
    const int64_t n = cols[iy].size();
    std::vector<const T*> ptrs;
    for (size_t i = iy; i < n_in; ++i) ptrs.push_back(cols[i].data());
    std::vector<T> b(pp), s(pp), t(pp), p(pp), lo(pp), hi(pp);
    typename Api<T>::Report rep;
    rep.beta = b.data(); rep.std_err = s.data(); rep.t = t.data(); rep.p = p.data(); rep.ci_lower = lo.data(); rep.ci_upper = hi.data();
    if (any_null) {
        std::vector<const uint8_t*> bms;
        std::vector<int64_t> offs;
        for (size_t i = iy; i < n_in; ++i) {
            bms.push_back(cols[i].validity.empty() ? nullptr : cols[i].validity.data());
            offs.push_back(0);
        }
        const int code = pol.kind == Policy::FILL ? PDS_NULL_FILL : pol.kind == Policy::SKIP ? PDS_NULL_SKIP : PDS_NULL_IGNORE;
        int64_t n_used = 0;
        check(Api<T>::report_nullable(thread_ctx(), ptrs.data(), bms.data(), offs.data(), n_feat, n, PDS_HOST, code, (T)pol.fill, bias,
                                      se_type, y_var, &rep, &n_used));
    } else {
        check(Api<T>::report(thread_ctx(), ptrs.data(), wls ? cols[0].data() : nullptr, n_feat, n, PDS_HOST, bias, se_type, y_var,
                             &rep));
    }
    std::vector<std::string> names;
    for (size_t i = iy + 1; i < n_in; ++i) names.push_back(cols[i].name);
    if (bias) names.push_back("__bias__");
    std::vector<T> r2(pp, rep.r2), ar2(pp, rep.adj_r2);
    const char* se_name = wls ? "std_err" : (se_type == PDS_HC0 ? "hc0_se" : se_type == PDS_HC1 ? "hc1_se" : se_type == PDS_HC2 ? "hc2_se"
                                             : se_type == PDS_HC3 ? "hc3_se" : "std_err");
    const char* fnames[9] = {"features", "beta", se_name, "t", "p>|t|", "0.025", "0.975", "r2", "adj_r2"};
    std::vector<std::unique_ptr<ArrowArray>> kids;
    kids.push_back(utf8_array(names));
    for (auto* v : {&b, &s, &t, &p, &lo, &hi, &r2, &ar2}) kids.push_back(prim_array<T>(v->data(), pp, nullptr));
    std::vector<std::unique_ptr<ArrowSchema>> sk;
    sk.push_back(make_schema("U", fnames[0]));
    for (int i = 1; i < 9; ++i) sk.push_back(make_schema(fmt_of<T>(), fnames[i]));
    export_series(out, make_schema("+s", "lin_reg_report", std::move(sk)), struct_array(pp, std::move(kids)));
}

// ------------------------------------------------------------------------------------------------- rolling / recursive
template <typename T>
void do_windowed(SeriesExport* in, size_t n_in, const Kwargs& kw, SeriesExport* out, bool rolling) {
    auto cols = import_all<T>(in, n_in);
    if (n_in < 2) raise("need a target and at least one feature");
    const bool bias = kw_bool(kw, "bias");
    const int64_t nwin = kw_i64(kw, "n");
    const double lambda = kw_f64(kw, "lambda");
    int64_t min_size = kw_i64(kw, "min_size");
    Policy pol = parse_policy(kw_str(kw, "null_policy", "raise"));
    const int n_feat = (int)n_in - 1;
    const int pp = n_feat + (bias ? 1 : 0);
    const int64_t n = cols[0].size();
    bool has_null = false;
    for (auto& c : cols) has_null |= c.null_count > 0;
    bool skipping = false;
    std::vector<uint8_t> keep;  // recursive + skip / fill: rows that survive the policy (the fit runs on the compacted frame)
    if (has_null && !rolling && pol.kind != Policy::RAISE && pol.kind != Policy::IGNORE) {
        // pl_recursive_lr :1131-1181: series_to_mat_for_lr drops the rows (skip: any null; fill: null target, feature nulls
        // take the fill value), the expanding fit runs on what is left and the result is spread back: null until `n` kept
        // rows have been seen, null on every dropped row.  (The reference forms pred from compacted row i instead of the
        // row the coefficients belong to, :1161-1162 -- not replicated: pred is x_row . coeffs_row as on null-free data.)
        keep.assign(n, 1);
        for (size_t ci = 0; ci < cols.size(); ++ci) {
            auto& c = cols[ci];
            if (!c.null_count) continue;
            const bool drop = pol.kind != Policy::FILL || ci == 0;
            for (int64_t i = 0; i < n; ++i)
                if (!bit_get(c.validity.data(), i)) {
                    if (drop) keep[i] = 0;
                    else c.own()[i] = (T)pol.fill;
                }
        }
        int64_t nk = 0;
        for (int64_t i = 0; i < n; ++i) nk += keep[i];
        if (nk == 0) raise("Empty data");
        for (auto& c : cols) {
            std::vector<T>& v = c.own();
            int64_t j = 0;
            for (int64_t i = 0; i < n; ++i)
                if (keep[i]) v[j++] = v[i];
            c.shrink(nk);
        }
        has_null = false;
    }
    if (has_null) {
        if (pol.kind == Policy::RAISE) raise("Nulls found in data");
        // rolling: SKIP -> SKIP_WINDOW, FILL -> FILL_WINDOW (:1215-1219): data stays in place, nulls become NaN (features
        // under FILL: the fill value) and the window algorithm leaves non-finite rows out
        for (size_t ci = 0; ci < cols.size(); ++ci) {
            auto& c = cols[ci];
            if (!c.null_count) continue;
            const T repl = (pol.kind == Policy::FILL && ci > 0) ? (T)pol.fill : (T)NAN;
            std::vector<T>& v = c.own();
            for (int64_t i = 0; i < n; ++i)
                if (!bit_get(c.validity.data(), i)) v[i] = repl;
        }
        skipping = rolling && (pol.kind == Policy::SKIP || pol.kind == Policy::SKIP_WINDOW || pol.kind == Policy::FILL);
    }
    if (!skipping) min_size = 0;
    std::vector<const T*> ptrs;
    for (auto& c : cols) ptrs.push_back(c.data());
    // the copies back from the device land in the storage the result arrays will own (7.2 GB of coefficients at 1e8 x 8)
    ByteVec cbuf = raw_buffer<T>((size_t)n * pp), pbuf = raw_buffer<T>((size_t)n);
    T* const coeffs = as<T>(cbuf);
    T* const pred = as<T>(pbuf);
    std::vector<uint8_t> valid(n);
    if (rolling)
        check(Api<T>::rolling(thread_ctx(), ptrs.data(), n_feat, n, PDS_HOST, bias, nwin, min_size, (T)lambda, coeffs, pred, valid.data()));
    else if (keep.empty())
        check(Api<T>::recursive(thread_ctx(), ptrs.data(), n_feat, n, PDS_HOST, bias, nwin, (T)lambda, coeffs, pred, valid.data()));
    else {
        const int64_t nk = cols[0].size();
        std::fill(valid.begin(), valid.end(), (uint8_t)0);
        if (nk >= nwin && nk >= pp) {  // fewer kept rows than the initial fit needs: every row stays null
            std::vector<T> cc((size_t)nk * pp), pc(nk);
            std::vector<uint8_t> vc(nk);
            check(Api<T>::recursive(thread_ctx(), ptrs.data(), n_feat, nk, PDS_HOST, bias, nwin, (T)lambda, cc.data(), pc.data(), vc.data()));
            if (reference_quirks())  // pred of fitted row j from compacted row j - (n - 1)
                for (int64_t j = nwin - 1; j < nk; ++j) {
                    const int64_t r = j - (nwin - 1);
                    T acc = bias ? cc[(size_t)j * pp + n_feat] : (T)0;
                    for (int k = 0; k < n_feat; ++k) acc += ptrs[1 + k][r] * cc[(size_t)j * pp + k];
                    pc[j] = acc;
                }
            int64_t j = 0;
            for (int64_t i = 0; i < n; ++i) {
                if (!keep[i]) continue;
                std::copy(cc.begin() + j * pp, cc.begin() + (j + 1) * pp, coeffs + i * pp);
                pred[i] = pc[j];
                valid[i] = vc[j];
                ++j;
            }
        }
    }
    // Struct{coeffs: List<T>[p'], pred: T}; invalid rows are null in both fields (:1241-1269)
    std::vector<std::unique_ptr<ArrowArray>> kids;
    kids.push_back(list_array_take_rows<T>(std::move(cbuf), n, pp, valid.data()));
    kids.push_back(prim_array_take<T>(std::move(pbuf), n, valid.data()));
    std::vector<std::unique_ptr<ArrowSchema>> sk;
    sk.push_back(list_schema<T>("coeffs"));
    sk.push_back(make_schema(fmt_of<T>(), "pred"));
    export_series(out, make_schema("+s", "", std::move(sk)), struct_array(n, std::move(kids)));
}

// ------------------------------------------------------------------------------------------------- pl_lr_by / pl_lr_by_pred (new)
// inputs: [key (integer, any row order, nulls = one group), weights?, y, x1..xp]
//   pl_lr_by       Struct{key, coeffs: List<T>}, one row per group, keys ascending (a null key's group where its stand-in sorts)
//   pl_lr_by_pred  Struct{pred, resid}, one row per INPUT row, in the frame's row order: what
//                  `group_by(key).agg(lin_reg(..., return_pred=True))` / `lin_reg(..., return_pred=True).over(key)` compute per
//                  group (linear_regression.rs:704-820); rows of a null group are null in both fields (:745-750)
template <typename T>
void do_lr_by(SeriesExport* in, size_t n_in, const Kwargs& kw, SeriesExport* out, bool want_pred) {
    const bool weighted = kw_bool(kw, "weighted");
    if (n_in < (weighted ? 4u : 3u)) raise("pl_lr_by needs a key, a target and at least one feature");
    const pds_lr_params prm = lr_params(kw);
    auto key = import_series<int64_t>(in[0]);  // an Int64 key column is borrowed as it is; other integer widths are widened
    std::vector<Column<T>> cols;
    for (size_t i = 1; i < n_in; ++i) cols.push_back(import_series<T>(in[i]));
    const Policy pol = parse_policy(kw_str(kw, "null_policy", "raise"));
    bool any_null = false;
    for (auto& c : cols) any_null |= c.null_count > 0;
    if (any_null && pol.kind == Policy::RAISE) raise("Nulls found in data");
    // pl_lr never compacts its weights (:436-446): a weighted fit on a frame that loses rows fails in the reference too
    if (any_null && weighted) raise("Shape of weights is not the same as the data.");
    int64_t n = key.size();
    for (auto& c : cols)
        if (c.size() != n) raise("input columns differ in length");
    // per-row predictions of a frame with nulls: what every group's pl_lr_pred does with them (linear_regression.rs:151-267,
    // 790-812) -- "skip" fits on the rows without a null, a fill policy fills the features and drops the rows whose target is
    // null; the dropped rows come back as null rows.  The frame is compacted on the host, fitted, and re-expanded below.
    std::vector<uint8_t> keep;   // non-empty: the call runs on the kept rows only
    const int64_t n_full = n;
    // "ignore": a group that holds a null fits on NaN (series_to_mat_for_lr keeps the rows, null -> NaN, linear_regression.rs:196,
    // utils/mod.rs:146-154) and every one of ITS rows comes back NaN (valid, not null -- this library's answer for pl_lr_pred + ignore,
    // DESIGN 7); the other groups are fitted as usual.  The nulls are zeroed for the fit and the rows of the affected keys overwritten.
    std::vector<uint8_t> row_has_null;
    if (any_null && want_pred && pol.kind == Policy::IGNORE) {
        row_has_null.assign((size_t)n, 0);
        for (auto& c : cols) {
            if (!c.null_count) continue;
            auto& v = c.own();
            for (int64_t i = 0; i < n; ++i)
                if (!bit_get(c.validity.data(), i)) {
                    v[i] = T(0);
                    row_has_null[i] = 1;
                }
            c.validity.clear();
            c.null_count = 0;
        }
        any_null = false;
    }
    if (any_null && want_pred) {
        if (pol.kind != Policy::SKIP && pol.kind != Policy::FILL)
            raise("pl_lr_by_pred: rows with nulls take null_policy 'skip', 'ignore' or a fill value");
        if (pol.kind == Policy::FILL)
            for (size_t k = 1; k < cols.size(); ++k) {
                auto& c = cols[k];
                if (!c.null_count) continue;
                auto& v = c.own();
                for (int64_t i = 0; i < n; ++i)
                    if (!bit_get(c.validity.data(), i)) v[i] = (T)pol.fill;
            }
        const size_t judged = pol.kind == Policy::FILL ? 1 : cols.size();
        keep.assign((size_t)n, 1);
        for (size_t k = 0; k < judged; ++k)
            if (cols[k].null_count)
                for (int64_t i = 0; i < n; ++i)
                    if (!bit_get(cols[k].validity.data(), i)) keep[i] = 0;
        int64_t m = 0;
        for (int64_t i = 0; i < n; ++i) m += keep[i];
        if (m == n) {
            keep.clear();  // (only filled features: nothing was dropped)
        } else {
            for (auto& c : cols) {
                auto& v = c.own();
                int64_t d = 0;
                for (int64_t i = 0; i < n; ++i)
                    if (keep[i]) v[d++] = v[i];
                c.shrink(m);
            }
            auto& kv = key.own();
            int64_t d = 0;
            std::vector<uint8_t> kvalid;
            if (key.null_count > 0) kvalid.assign((size_t)(m + 7) / 8, 0);
            for (int64_t i = 0; i < n; ++i)
                if (keep[i]) {
                    if (key.null_count > 0 && bit_get(key.validity.data(), i)) bit_set(kvalid, d);
                    kv[d++] = kv[i];
                }
            key.shrink(m);
            if (key.null_count > 0) {
                key.validity.swap(kvalid);
                int64_t nn = 0;
                for (int64_t i = 0; i < m; ++i) nn += bit_get(key.validity.data(), i) ? 0 : 1;
                key.null_count = nn;
                if (nn == 0) key.validity.clear();
            }
            n = m;
        }
        for (auto& c : cols) {
            c.validity.clear();
            c.null_count = 0;
        }
        any_null = false;
        if (n == 0) raise("Empty data");
    }
    // Polars' group_by makes the null keys ONE group.  They take a key value no valid row uses -- max + 1, so that group comes
    // last (min - 1 when the maximum is INT64_MAX) -- and the group that carries it is reported with a null key.
    bool null_group = false;
    int64_t null_stand_in = 0;
    if (key.null_count > 0) {
        const int64_t* k = key.data();
        int64_t mx = std::numeric_limits<int64_t>::min(), mn = std::numeric_limits<int64_t>::max();
        bool any_valid = false;
        for (int64_t i = 0; i < n; ++i)
            if (bit_get(key.validity.data(), i)) {
                mx = std::max(mx, k[i]);
                mn = std::min(mn, k[i]);
                any_valid = true;
            }
        if (!any_valid) null_stand_in = 0;
        else if (mx < std::numeric_limits<int64_t>::max()) null_stand_in = mx + 1;
        else if (mn > std::numeric_limits<int64_t>::min()) null_stand_in = mn - 1;
        else raise("pl_lr_by: the keys span the whole int64 range, no value is left for the null group");
        auto& kv = key.own();
        for (int64_t i = 0; i < n; ++i)
            if (!bit_get(key.validity.data(), i)) kv[i] = null_stand_in;
        null_group = true;
    }
    const int n_feat = (int)n_in - 2 - (weighted ? 1 : 0);
    const int pp = n_feat + prm.add_bias;
    const int64_t* ikey = key.data();
    if (n == 0) raise("Empty data");
    // result storage is sized for a guessed capacity but never zeroed: pages the device-to-host copies do not write are
    // never touched (a zeroed 1M-group guess cost 16 ms of page faults per call)
    RawVec<int64_t> keys;
    ByteVec cobuf;
    RawVec<uint8_t> nulls;
    ByteVec pred_b, resid_b;
    RawVec<uint8_t> row_null;
    int64_t ng = 0;
    if (!any_null) {
        // keys in any row order: the device brings the frame into key order (radix sort + gather; the weights ride along as
        // one more column) -- the grouping Polars' group_by does on the host before it calls pl_lr once per group
        const size_t first = weighted ? 1 : 0;
        std::vector<const T*> ptrs;
        for (size_t c = first; c < cols.size(); ++c) ptrs.push_back(cols[c].data());
        const T* wts = weighted ? cols[0].data() : nullptr;
        if (want_pred) {
            pred_b = raw_buffer<T>((size_t)n);
            resid_b = raw_buffer<T>((size_t)n);
            row_null.resize(n);
            // large frames: the sliced route (as for the coefficient fit below), slices independent
            const int64_t multi_min = settings().by_key_multi_min_rows;
            std::unique_lock<std::mutex> multi(MultiContexts::get().busy, std::defer_lock);
            if (n >= multi_min && multi_min > 0 && multi.try_lock() && MultiContexts::get().contexts().size() > 1) {
                const auto& cx = MultiContexts::get().contexts();
                check(Api<T>::by_key_pred_multi(cx.data(), (int)cx.size(), settings().by_key_slices, ptrs.data(), wts, ikey, n_feat, n, &prm,
                                                as<T>(pred_b), as<T>(resid_b), row_null.data()));
            } else {
                check(Api<T>::by_key_pred(thread_ctx(), ptrs.data(), wts, ikey, n_feat, n, PDS_HOST, &prm, n, nullptr, nullptr, nullptr,
                                          nullptr, as<T>(pred_b), as<T>(resid_b), row_null.data()));
            }
        } else {
            // output capacity: the number of distinct keys is unknown until the device has counted the runs -- start from a
            // guess (every row its own group is always enough but costs n x p' of host memory) and repeat with the count once
            int64_t cap = n <= ((int64_t)1 << 20) ? n : std::max<int64_t>((int64_t)1 << 20, n / 16);
            for (int attempt = 0;; ++attempt) {
                keys.resize(cap);
                cobuf = raw_buffer<T>((size_t)cap * pp);
                nulls.resize(cap);
                int rc;
                // large unweighted frames: the sliced route -- several contexts (devices: PDS_DEVICES; per device:
                // PDS_BY_KEY_CONTEXTS), every slice over its context's stream / its device's PCIe link (capi_multi.hpp); it
                // falls back to the single-context entry point by itself when the keys are not in order
                // (plugin_settings.hpp: PDS_BY_KEY_MULTI_MIN_ROWS=0 switches the route off, PDS_BY_KEY_SLICES sets the slice count)
                const int64_t multi_min = settings().by_key_multi_min_rows;
                const int slices = settings().by_key_slices;
                std::unique_lock<std::mutex> multi(MultiContexts::get().busy, std::defer_lock);
                if (weighted) {
                    rc = Api<T>::by_key_pred(thread_ctx(), ptrs.data(), wts, ikey, n_feat, n, PDS_HOST, &prm, cap, keys.data(), as<T>(cobuf),
                                             nulls.data(), &ng, nullptr, nullptr, nullptr);
                } else if (n >= multi_min && multi_min > 0 && multi.try_lock() && MultiContexts::get().contexts().size() > 1) {
                    const auto& cx = MultiContexts::get().contexts();
                    rc = Api<T>::by_key_multi(cx.data(), (int)cx.size(), slices, ptrs.data(), ikey, n_feat, n, &prm, cap, keys.data(),
                                              as<T>(cobuf), nulls.data(), &ng);
                } else {
                    rc = Api<T>::by_key(thread_ctx(), ptrs.data(), ikey, n_feat, n, PDS_HOST, &prm, cap, keys.data(), as<T>(cobuf),
                                        nulls.data(), &ng);
                }
                if (rc != 0 && attempt == 0 && ng > cap) {
                    cap = ng;
                    continue;
                }
                check(rc);
                break;
            }
            keys.resize(ng);
            nulls.resize(ng);
        }
    } else {
        // rows with nulls: the rows are put into key order on the host (a stable sort of the row indices; the validity bitmaps
        // travel with their rows), then the bitmap-aware grouped entry point applies the policy inside every group
        bool ordered = true;
        for (int64_t i = 1; i < n && ordered; ++i) ordered = ikey[i] >= ikey[i - 1];
        if (!ordered) {
            std::vector<int64_t> perm(n);
            for (int64_t i = 0; i < n; ++i) perm[i] = i;
            std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return ikey[a] < ikey[b]; });
            std::vector<int64_t> k2(n);
            for (int64_t i = 0; i < n; ++i) k2[i] = ikey[perm[i]];
            key.view = nullptr;
            key.values.swap(k2);
            ikey = key.values.data();
            for (auto& c : cols) {
                std::vector<T> v2(n);
                const T* src = c.data();
                for (int64_t i = 0; i < n; ++i) v2[i] = src[perm[i]];
                c.view = nullptr;
                c.values.swap(v2);
                if (!c.validity.empty()) {
                    std::vector<uint8_t> b2(c.validity.size(), 0);
                    for (int64_t i = 0; i < n; ++i)
                        if (bit_get(c.validity.data(), perm[i])) bit_set(b2, i);
                    c.validity.swap(b2);
                }
            }
        }
        std::vector<int64_t> off = {0};
        for (int64_t i = 0; i < n; ++i)
            if (i == 0 || ikey[i] != ikey[i - 1]) {
                if (i) off.push_back(i);
                keys.push_back(ikey[i]);
            }
        off.push_back(n);
        ng = (int64_t)keys.size();
        std::vector<const T*> ptrs;
        for (auto& c : cols) ptrs.push_back(c.data());
        cobuf = raw_buffer<T>((size_t)ng * pp);
        nulls.resize(ng);
        std::vector<const uint8_t*> bms;
        std::vector<int64_t> offs;
        for (auto& c : cols) {
            bms.push_back(c.validity.empty() ? nullptr : c.validity.data());
            offs.push_back(0);
        }
        const int code = pol.kind == Policy::FILL ? PDS_NULL_FILL : pol.kind == Policy::SKIP ? PDS_NULL_SKIP : PDS_NULL_IGNORE;
        check(Api<T>::grouped_nullable(thread_ctx(), ptrs.data(), bms.data(), offs.data(), n_feat, n, off.data(), ng, PDS_HOST, code,
                                       (T)pol.fill, &prm, as<T>(cobuf), nulls.data()));
    }
    if (want_pred && !keep.empty()) {
        // re-expansion through the row mask (linear_regression.rs:790-812): dropped rows are null rows of the result
        ByteVec pf = raw_buffer<T>((size_t)n_full), rf = raw_buffer<T>((size_t)n_full);
        std::vector<uint8_t> valid((size_t)n_full, 0);
        T* pd = as<T>(pf);
        T* rd = as<T>(rf);
        const T* ps = as<T>(pred_b);
        const T* rs = as<T>(resid_b);
        const T nanv = std::numeric_limits<T>::quiet_NaN();
        int64_t k = 0;
        for (int64_t i = 0; i < n_full; ++i) {
            if (keep[i]) {
                pd[i] = ps[k];
                rd[i] = rs[k];
                valid[i] = row_null[k] ? 0 : 1;
                ++k;
            } else {
                pd[i] = nanv;
                rd[i] = nanv;
            }
        }
        std::vector<std::unique_ptr<ArrowArray>> kids;
        kids.push_back(prim_array_take<T>(std::move(pf), n_full, valid.data()));
        kids.push_back(prim_array_take<T>(std::move(rf), n_full, valid.data()));
        std::vector<std::unique_ptr<ArrowSchema>> sk;
        sk.push_back(make_schema(fmt_of<T>(), "pred"));
        sk.push_back(make_schema(fmt_of<T>(), "resid"));
        export_series(out, make_schema("+s", "", std::move(sk)), struct_array(n_full, std::move(kids)));
        return;
    }
    if (want_pred && !row_has_null.empty()) {
        std::unordered_set<int64_t> bad;  // (the null key's rows carry its stand-in by now)
        for (int64_t i = 0; i < n; ++i)
            if (row_has_null[i]) bad.insert(ikey[i]);
        T* pd = as<T>(pred_b);
        T* rd = as<T>(resid_b);
        const T nanv = std::numeric_limits<T>::quiet_NaN();
        for (int64_t i = 0; i < n; ++i)
            if (bad.count(ikey[i])) {
                pd[i] = nanv;
                rd[i] = nanv;
                row_null[i] = 0;
            }
    }
    if (want_pred) {
        // validity of the two children: none at all when no row is null (the usual case: a scan of the n flag bytes instead of
        // a second n-byte vector and two packing passes -- 90 ms of the headline frame's 370); otherwise packed once, copied once
        std::vector<std::unique_ptr<ArrowArray>> kids;
        if (!any_flag(row_null.data(), n)) {
            kids.push_back(prim_array_take<T>(std::move(pred_b), n, nullptr));
            kids.push_back(prim_array_take<T>(std::move(resid_b), n, nullptr));
        } else {
            for (int64_t i = 0; i < n; ++i) row_null[i] = row_null[i] ? 0 : 1;  // (in place: null flags -> valid flags)
            ByteVec bm;
            const int64_t nulls = pack_validity(row_null.data(), n, bm);
            ByteVec bm2(bm);
            kids.push_back(prim_array_take_bitmap<T>(std::move(pred_b), n, std::move(bm), nulls));
            kids.push_back(prim_array_take_bitmap<T>(std::move(resid_b), n, std::move(bm2), nulls));
        }
        std::vector<std::unique_ptr<ArrowSchema>> sk;
        sk.push_back(make_schema(fmt_of<T>(), "pred"));
        sk.push_back(make_schema(fmt_of<T>(), "resid"));
        export_series(out, make_schema("+s", "", std::move(sk)), struct_array(n, std::move(kids)));
        return;
    }
    std::vector<uint8_t> ok(ng);
    for (int64_t g = 0; g < ng; ++g) ok[g] = nulls[g] ? 0 : 1;
    std::vector<std::unique_ptr<ArrowArray>> kids;
    {
        std::vector<uint8_t> kvalid;
        if (null_group) {
            kvalid.assign(ng, 1);
            for (int64_t g = 0; g < ng; ++g)
                if (keys[g] == null_stand_in) kvalid[g] = 0;
        }
        kids.push_back(prim_array_take<int64_t>(bytes_of(keys.data(), keys.size()), ng, null_group ? kvalid.data() : nullptr));
    }
    kids.push_back(list_array_take_rows<T>(std::move(cobuf), ng, pp, ok.data()));
    std::vector<std::unique_ptr<ArrowSchema>> sk;
    sk.push_back(make_schema("l", key.name.empty() ? "key" : key.name));
    sk.push_back(list_schema<T>("coeffs"));
    export_series(out, make_schema("+s", "", std::move(sk)), struct_array(ng, std::move(kids)));
}

// ------------------------------------------------------------------------------------------------- pl_lr_multi(_pred)
template <typename T>
void do_lr_multi(SeriesExport* in, size_t n_in, const Kwargs& kw, SeriesExport* out, bool want_pred) {
    // MultiLRKwargs :47-56; inputs [t_0 .. t_{k-1}, x_1 .. x_p]; series_to_mat_for_multi_lr :270-349
    const int k = (int)kw_i64(kw, "last_target_idx");
    if (k < 1 || (size_t)k >= n_in) raise("multi-target lin_reg needs targets and at least one feature");
    const bool bias = kw_bool(kw, "bias");
    const std::string sv = kw_str(kw, "solver", "qr");
    const int solver = sv == "svd" ? PDS_SOLVER_SVD : (sv == "choleskey" ? PDS_SOLVER_CHOLESKEY : PDS_SOLVER_QR);
    const Policy pol = parse_policy(kw_str(kw, "null_policy", "raise"));
    auto cols = import_all<T>(in, n_in);
    const int64_t n = cols[0].size();
    bool y_null = false, any_null = false;
    for (size_t i = 0; i < n_in; ++i) {
        any_null |= cols[i].null_count > 0;
        if ((int)i < k) y_null |= cols[i].null_count > 0;
    }
    if (any_null) {
        if (pol.kind == Policy::RAISE) raise("Nulls found in data");
        if (pol.kind != Policy::FILL) raise("The null policy is not supported by multi-target linear regression.");
        if (y_null) raise("Filling null doesn't work for multi-target lstsq when there are nulls in any of the targets.");
        for (size_t i = k; i < n_in; ++i)
            if (cols[i].null_count)
                for (int64_t r = 0; r < n; ++r)
                    if (!bit_get(cols[i].validity.data(), r)) cols[i].own()[r] = (T)pol.fill;
    }
    const int n_feat = (int)n_in - k;
    const int pp = n_feat + (bias ? 1 : 0);
    std::vector<const T*> ptrs;
    for (auto& c : cols) ptrs.push_back(c.data());
    std::vector<T> coeffs((size_t)k * pp), pred, resid;
    if (want_pred) {
        pred.resize((size_t)k * n);
        resid.resize((size_t)k * n);
    }
    int is_null = 0;
    check(Api<T>::multi(thread_ctx(), ptrs.data(), k, n_feat, n, PDS_HOST, bias, (T)kw_f64(kw, "l2_reg"), solver,
                        (T)kw_f64(kw, "singular_x_tol"), coeffs.data(), &is_null, want_pred ? pred.data() : nullptr,
                        want_pred ? resid.data() : nullptr));
    std::vector<std::unique_ptr<ArrowArray>> kids;
    std::vector<std::unique_ptr<ArrowSchema>> sk;
    if (!want_pred) {
        for (int t = 0; t < k; ++t) {
            const uint8_t ok = is_null ? 0 : 1;
            std::vector<int64_t> off = {0, is_null ? 0 : (int64_t)pp};
            kids.push_back(list_array<T>(off, &ok, coeffs.data() + (size_t)t * pp, is_null ? 0 : pp));
            sk.push_back(list_schema<T>(cols[t].name));
        }
        export_series(out, make_schema("+s", "coeffs", std::move(sk)), struct_array(1, std::move(kids)));
        return;
    }
    std::vector<uint8_t> valid(n, is_null ? 0 : 1);  // null_multi_pred :399-417 when gated
    for (int t = 0; t < k; ++t) {
        kids.push_back(prim_array<T>(pred.data() + (size_t)t * n, n, valid.data()));
        kids.push_back(prim_array<T>(resid.data() + (size_t)t * n, n, valid.data()));
        sk.push_back(make_schema(fmt_of<T>(), cols[t].name + "_pred"));
        sk.push_back(make_schema(fmt_of<T>(), cols[t].name + "_resid"));
    }
    export_series(out, make_schema("+s", "all_preds", std::move(sk)), struct_array(n, std::move(kids)));
}

// ------------------------------------------------------------------------------------------------- pl_lr_w_rcond
// series_to_mat_for_lr's null policies (linear_regression.rs:187-248) applied on the host to owned copies of the columns
// [y, x1..xp]; only the entry points without a device-side nullable form come here (pl_lr_w_rcond), and only when a bitmap
// actually holds a null.
template <typename T>
void host_null_policy(std::vector<Column<T>>& cols, const Policy& pol) {
    bool any = false;
    for (auto& c : cols) any |= c.null_count > 0;
    if (!any) return;
    const int64_t n = cols[0].size();
    if (pol.kind == Policy::RAISE) raise("Nulls found in data");
    const T nan = std::numeric_limits<T>::quiet_NaN();
    auto is_null = [&](const Column<T>& c, int64_t i) { return c.null_count > 0 && !bit_get(c.validity.data(), i); };
    if (pol.kind == Policy::IGNORE || pol.kind == Policy::SKIP_WINDOW) {  // keep rows, null -> NaN (utils/mod.rs:146-154)
        for (auto& c : cols) {
            if (!c.null_count) continue;
            auto& v = c.own();
            for (int64_t i = 0; i < n; ++i)
                if (!bit_get(c.validity.data(), i)) v[i] = nan;
        }
    } else {
        if (pol.kind == Policy::FILL)  // features filled, rows with a null target dropped
            for (size_t k = 1; k < cols.size(); ++k) {
                auto& c = cols[k];
                if (!c.null_count) continue;
                auto& v = c.own();
                for (int64_t i = 0; i < n; ++i)
                    if (!bit_get(c.validity.data(), i)) v[i] = (T)pol.fill;
            }
        const size_t judged = pol.kind == Policy::FILL ? 1 : cols.size();
        std::vector<uint8_t> keep((size_t)n, 1);
        bool drop = false;
        for (size_t k = 0; k < judged; ++k)
            if (cols[k].null_count)
                for (int64_t i = 0; i < n; ++i)
                    if (is_null(cols[k], i)) {
                        keep[i] = 0;
                        drop = true;
                    }
        if (drop)
            for (auto& c : cols) {
                auto& v = c.own();
                int64_t m = 0;
                for (int64_t i = 0; i < n; ++i)
                    if (keep[i]) v[m++] = v[i];
                c.shrink(m);
            }
    }
    for (auto& c : cols) {
        c.validity.clear();
        c.null_count = 0;
    }
}

template <typename T>
void do_lr_rcond(SeriesExport* in, size_t n_in, const Kwargs& kw, SeriesExport* out) {
    // pl_lr_w_rcond :651-702 / pl_lr_w_rcond_f32 linear_regression_f32.rs:515-566: Struct{coeffs: List, singular_values: List}
    const Policy pol = parse_policy(kw_str(kw, "null_policy", "raise"));
    auto cols = import_all<T>(in, n_in);
    if (n_in < 2) raise("need a target and at least one feature");
    for (auto& c : cols)
        if (c.size() != cols[0].size()) raise("input columns differ in length");
    if (cols[0].size() == 0) raise("Empty data");
    host_null_policy<T>(cols, pol);
    const bool bias = kw_bool(kw, "bias");
    const int n_feat = (int)n_in - 1;
    const int pp = n_feat + (bias ? 1 : 0);
    const int64_t n = cols[0].size();
    // rcond rides in as `tol`: (kwargs.tol as T).max(T::EPSILON * max(nrows, p') as T)
    const T rcond = std::max((T)kw_f64(kw, "tol"), std::numeric_limits<T>::epsilon() * (T)std::max<int64_t>(n, pp));
    std::vector<const T*> ptrs;
    for (auto& c : cols) ptrs.push_back(c.data());
    std::vector<T> b(pp), sv(pp);
    check(Api<T>::rcond(thread_ctx(), ptrs.data(), n_feat, n, PDS_HOST, bias, (T)kw_f64(kw, "l2_reg"), rcond, b.data(), sv.data()));
    const uint8_t ok = 1;
    std::vector<int64_t> off = {0, (int64_t)pp};
    std::vector<std::unique_ptr<ArrowArray>> kids;
    kids.push_back(list_array<T>(off, &ok, b.data(), pp));
    kids.push_back(list_array<T>(off, &ok, sv.data(), pp));
    std::vector<std::unique_ptr<ArrowSchema>> sk;
    sk.push_back(list_schema<T>("coeffs"));
    sk.push_back(list_schema<T>("singular_values"));
    export_series(out, make_schema("+s", "", std::move(sk)), struct_array(1, std::move(kids)));
}

template <typename F>
void guarded(SeriesExport* ret, F&& f) {
    try {
        g_plugin_err.clear();
        f();
    } catch (const PluginError& e) {
        g_plugin_err = e.msg;
        if (ret) std::memset(ret, 0, sizeof(*ret));
    } catch (const std::exception& e) {
        g_plugin_err = std::string("PANIC: ") + e.what();
        if (ret) std::memset(ret, 0, sizeof(*ret));
    }
}
