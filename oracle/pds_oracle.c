/*
 * oracle/pds_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's least-squares expression path (polars_ds v0.12.1):
 *   src/linear/lr/lr_solvers.rs, src/linear/online_lr/lr_online_solvers.rs,
 *   src/num_ext/linear_regression.rs (report arithmetic), src/stats_utils/{beta,gamma}.rs.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The shipped product path (polars_ds_extension_amd/csrc) never links or calls it.
 *
 * Parity status: the reference crate cannot be built in this image (no Rust toolchain) and its dense
 * arithmetic lives in faer 0.23.2 (git rev 8377404e78, not vendored).  This restatement is pinned
 * (tests/test_oracle_*.py) against (1) the literal outputs printed in the reference's
 * examples/basics.ipynb (6 significant digits), (2) the literal frames in the reference's tests,
 * (3) numpy / scipy / scikit-learn on the reference's seeded fixtures with the reference's own
 * tolerances.  Bit-level parity with faer: UNPINNED.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off so the special functions keep the reference's
 * operation order and rounding).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "pds_oracle.h"

#define ORC_MAX_P 1024

/* ========================================================================================== */
/* src/stats_utils restated (f64 only, like the reference)                                     */
/* ========================================================================================== */

static const double PREC_ACC = 0.0000000000000011102230246251565; /* stats_utils/mod.rs:21 */
static const double LN_PI = 1.1447298858494001741434273513530587116472948129153;
static const double LN_2_SQRT_E_OVER_PI = 0.6207822376352452223455184457816472122518527279025978;
static const double GAMMA_R = 10.900511; /* gamma.rs:5 */
static const double GAMMA_DK[11] = {     /* gamma.rs:7-19 */
    2.48574089138753565546e-5, 1.05142378581721974210,    -3.45687097222016235469,
    4.51227709466894823700,    -2.98285225323576655721,   1.05639711577126713077,
    -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
    4.63399473359905636708e-6, -2.71994908488607703910e-9};

/* ref: gamma.rs:51-73 */
double orc_ln_gamma(double x) {
    const double E = 2.71828182845904523536028747135266250;
    const double PI = 3.14159265358979323846264338327950288;
    if (x < 0.5) {
        double s = GAMMA_DK[0];
        for (int i = 1; i < 11; ++i) s = s + GAMMA_DK[i] / ((double)i - x);
        return LN_PI - log(sin(PI * x)) - log(s) - LN_2_SQRT_E_OVER_PI -
               (0.5 - x) * log((0.5 - x + GAMMA_R) / E);
    } else {
        double s = GAMMA_DK[0];
        for (int i = 1; i < 11; ++i) s = s + GAMMA_DK[i] / (x + (double)i - 1.0);
        return log(s) + LN_2_SQRT_E_OVER_PI + (x - 0.5) * log((x - 0.5 + GAMMA_R) / E);
    }
}

/* ref: beta.rs:62-157 (checked_beta_reg). *err = 1 on a domain error. */
double orc_beta_reg(double a, double b, double x, int* err) {
    if (err) *err = 0;
    if (a <= 0.0 || b <= 0.0 || !(x >= 0.0 && x <= 1.0)) {
        if (err) *err = 1;
        return NAN;
    }
    double bt = (x == 0.0 || x == 1.0)
                    ? 0.0
                    : exp(orc_ln_gamma(a + b) - orc_ln_gamma(a) - orc_ln_gamma(b) + a * log(x) +
                          b * log(1.0 - x));
    int symm = x >= (a + 1.0) / (a + b + 2.0);
    double eps = PREC_ACC;
    double fpmin = DBL_MIN / eps;
    if (symm) {
        double t = a;
        a = b;
        b = t;
        x = 1.0 - x;
    }
    double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0;
    double d = 1.0 - qab * x / qap;
    if (fabs(d) < fpmin) d = fpmin;
    d = 1.0 / d;
    double h = d;
    for (int mi = 1; mi < 141; ++mi) {
        double m = (double)mi;
        double m2 = m * 2.0;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < fpmin) d = fpmin;
        c = 1.0 + aa / c;
        if (fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d;
        h = h * d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < fpmin) d = fpmin;
        c = 1.0 + aa / c;
        if (fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) <= eps) return symm ? 1.0 - bt * h / a : bt * h / a;
    }
    return symm ? 1.0 - bt * h / a : bt * h / a;
}

/* ref: beta.rs:24-37.  df = inf branch uses the normal sf (normal.rs:503-505) */
double orc_student_t_sf(double x, double df, int* err) {
    if (err) *err = 0;
    if (isinf(df)) return 0.5 * erfc(x / 1.41421356237309504880168872420969808);
    double h = df / (df + x * x);
    double ib = 0.5 * orc_beta_reg(df / 2.0, 0.5, h, err);
    return x <= 0.0 ? 1.0 - ib : ib;
}

static double ln_beta(double a, double b) { /* beta.rs:164-170 */
    if (a <= 0.0 || b <= 0.0) return NAN;
    return orc_ln_gamma(a) + orc_ln_gamma(b) - orc_ln_gamma(a + b);
}

/* ref: beta.rs:201-361 (AS 64 / AS 109 with Newton-Raphson) */
double orc_inv_beta_reg(double a, double b, double x) {
    double lnb = ln_beta(a, b);
    const int SAE = -30;
    const double FPU = 1e-30;
    if (x == 0.0) return 0.0;
    if (x == 1.0) return 1.0;
    double p, q;
    int flip = 0.5 < x;
    if (flip) {
        p = a;
        a = b;
        b = p;
        x = 1.0 - x;
    }
    p = sqrt(-log(x * x));
    q = p - (2.30753 + 0.27061 * p) / (1.0 + (0.99229 + 0.04481 * p) * p);
    if (1.0 < a && 1.0 < b) {
        double r = (q * q - 3.0) / 6.0;
        double s = 1.0 / (2.0 * a - 1.0);
        double t = 1.0 / (2.0 * b - 1.0);
        double h = 2.0 / (s + t);
        double w = q * sqrt(h + r) / h - (t - s) * (r + 5.0 / 6.0 - 2.0 / (3.0 * h));
        p = a / (a + b * exp(2.0 * w));
    } else {
        double t = 1.0 / (9.0 * b);
        t = 2.0 * b * pow(1.0 - t + q * sqrt(t), 3.0);
        if (t <= 0.0) {
            p = 1.0 - exp((log((1.0 - x) * b) + lnb) / b);
        } else {
            t = 2.0 * (2.0 * a + b - 1.0) / t;
            if (t <= 1.0)
                p = exp((log(x * a) + lnb) / a);
            else
                p = 1.0 - 2.0 / (t + 1.0);
        }
    }
    if (p < 0.0001) p = 0.0001; /* clamp :295 */
    if (p > 0.9999) p = 0.9999;
    int e = (int)(-5.0 / a / a - 1.0 / pow(x, 0.2) - 13.0);
    double acu = e > SAE ? pow(10.0, (double)e) : FPU; /* f64::powi(10.0, e) */
    double pnext = 0.0, qprev = 0.0, sq = 1.0, prev = 1.0;
    for (;;) { /* 'outer */
        int err = 0;
        q = orc_beta_reg(a, b, p, &err);
        q = (q - x) * exp(lnb + (1.0 - a) * log(p) + (1.0 - b) * log(1.0 - p));
        /* ORACLE GUARD (not in the reference): for a > ~7.09e6 (dof > ~1.42e7) the exp() above
         * overflows at the clamped start p = 0.9999, q = -inf, and the reference's step-halving
         * loop below (beta.rs:322-334) never terminates (sq = inf is never < prev; then 0*inf =
         * NaN).  The reference therefore hangs in lin_reg_report for N - p > ~1.42e7.  The oracle
         * reports NaN instead of spinning; see DESIGN.md "reference defects". */
        if (!isfinite(q)) return NAN;
        if (q * qprev <= 0.0) prev = sq > FPU ? sq : FPU;
        double g = 1.0;
        int done = 0;
        for (;;) {
            for (;;) {
                double adj = g * q;
                sq = adj * adj;
                if (sq < prev) {
                    pnext = p - adj;
                    if (pnext >= 0. && pnext <= 1.) break;
                }
                g /= 3.0;
            }
            if (prev <= acu || q * q <= acu) {
                p = pnext;
                done = 1;
                break;
            }
            if (pnext != 0.0 && pnext != 1.0) break;
            g /= 3.0;
        }
        if (done) break;
        if (pnext == p) break;
        p = pnext;
        qprev = q;
    }
    return flip ? 1.0 - p : p;
}

/* ref: beta.rs:365-377 */
double orc_student_t_ppf(double x, double df) {
    double x1 = x >= 0.5 ? 1.0 - x : x;
    double a = 0.5 * df, b = 0.5;
    double y = orc_inv_beta_reg(a, b, 2.0 * x1);
    y = sqrt(df * (1. - y) / y);
    return x >= 0.5 ? y : -y;
}

/* ========================================================================================== */
/* type-generic solvers, instantiated for f64 and f32                                          */
/* ========================================================================================== */

#define REAL double
#define SUF(x) x##_f64
#define REAL_EPS DBL_EPSILON
#define RSQRT sqrt
#define RLOG log
#define REXP exp
#define RISFINITE(v) isfinite(v)
#define RSIGNBIT(v) signbit(v)
#include "pds_oracle_impl.inc"
#undef REAL
#undef SUF
#undef REAL_EPS
#undef RSQRT
#undef RLOG
#undef REXP
#undef RISFINITE
#undef RSIGNBIT

#define REAL float
#define SUF(x) x##_f32
#define REAL_EPS FLT_EPSILON
#define RSQRT sqrtf
#define RLOG logf
#define REXP expf
#define RISFINITE(v) isfinite(v)
#define RSIGNBIT(v) signbit(v)
#include "pds_oracle_impl.inc"
#undef REAL
#undef SUF
#undef REAL_EPS
#undef RSQRT
#undef RLOG
#undef REXP
#undef RISFINITE
#undef RSIGNBIT

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
