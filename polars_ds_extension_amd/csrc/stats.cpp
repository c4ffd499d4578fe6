// stats.cpp -- host-side special functions of the report path (p-values and the 97.5% t quantile).
//
// O(p) scalar work per regression, so it stays on the host, but parity with the reference at 1e-10 is
// only possible if the SAME algorithm is evaluated in the SAME operation order: at large dof the
// reference's own value is ~1e-7 away from the exact one (cancellation in lnG(a+b)-lnG(a)-lnG(b)).
// This file therefore restates /root/reference/src/stats_utils/{gamma.rs:51-73, beta.rs:24-37,62-157,
// 201-377}.  Compiled with -ffp-contract=off.
//
// One deliberate deviation: for dof > ~1.42e7 the reference's inv_beta_reg never terminates (its
// clamped start p=0.9999 makes exp() overflow and the step-halving loop spins on inf/NaN, beta.rs:311,
// 322-334).  student_t_ppf() detects that case and returns the Cornish-Fisher expansion of the
// quantile instead (error < 1e-15 for dof > 1e7), so lin_reg_report works at N = 1e8.
#include <cfloat>
#include <cmath>

#include "common.hpp"

namespace pds {
namespace {

const double PREC_ACC = 0.0000000000000011102230246251565;
const double LN_PI = 1.1447298858494001741434273513530587116472948129153;
const double LN_2_SQRT_E_OVER_PI = 0.6207822376352452223455184457816472122518527279025978;
const double GAMMA_R = 10.900511;
const double GAMMA_DK[11] = {2.48574089138753565546e-5,  1.05142378581721974210,    -3.45687097222016235469,
                             4.51227709466894823700,     -2.98285225323576655721,   1.05639711577126713077,
                             -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
                             4.63399473359905636708e-6,  -2.71994908488607703910e-9};

double ln_gamma(double x) {
    const double E = 2.71828182845904523536028747135266250;
    const double PI = 3.14159265358979323846264338327950288;
    if (x < 0.5) {
        double s = GAMMA_DK[0];
        for (int i = 1; i < 11; ++i) s = s + GAMMA_DK[i] / ((double)i - x);
        return LN_PI - std::log(std::sin(PI * x)) - std::log(s) - LN_2_SQRT_E_OVER_PI -
               (0.5 - x) * std::log((0.5 - x + GAMMA_R) / E);
    }
    double s = GAMMA_DK[0];
    for (int i = 1; i < 11; ++i) s = s + GAMMA_DK[i] / (x + (double)i - 1.0);
    return std::log(s) + LN_2_SQRT_E_OVER_PI + (x - 0.5) * std::log((x - 0.5 + GAMMA_R) / E);
}

double beta_reg(double a, double b, double x, bool* err) {
    if (a <= 0.0 || b <= 0.0 || !(x >= 0.0 && x <= 1.0)) {
        if (err) *err = true;
        return NAN;
    }
    const double bt = (x == 0.0 || x == 1.0)
                          ? 0.0
                          : std::exp(ln_gamma(a + b) - ln_gamma(a) - ln_gamma(b) + a * std::log(x) +
                                     b * std::log(1.0 - x));
    const bool symm = x >= (a + 1.0) / (a + b + 2.0);
    const double eps = PREC_ACC;
    const double fpmin = DBL_MIN / eps;
    if (symm) {
        std::swap(a, b);
        x = 1.0 - x;
    }
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0;
    double d = 1.0 - qab * x / qap;
    if (std::fabs(d) < fpmin) d = fpmin;
    d = 1.0 / d;
    double h = d;
    for (int mi = 1; mi < 141; ++mi) {
        const double m = (double)mi;
        const double m2 = m * 2.0;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d;
        if (std::fabs(d) < fpmin) d = fpmin;
        c = 1.0 + aa / c;
        if (std::fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d;
        h = h * d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d;
        if (std::fabs(d) < fpmin) d = fpmin;
        c = 1.0 + aa / c;
        if (std::fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (std::fabs(del - 1.0) <= eps) break;
    }
    return symm ? 1.0 - bt * h / a : bt * h / a;
}

// returns NaN when the reference's iteration would not terminate
double inv_beta_reg(double a, double b, double x) {
    const double lnb = (a <= 0.0 || b <= 0.0) ? NAN : ln_gamma(a) + ln_gamma(b) - ln_gamma(a + b);
    const int SAE = -30;
    const double FPU = 1e-30;
    if (x == 0.0) return 0.0;
    if (x == 1.0) return 1.0;
    double p, q;
    const bool flip = 0.5 < x;
    if (flip) {
        p = a;
        a = b;
        b = p;
        x = 1.0 - x;
    }
    p = std::sqrt(-std::log(x * x));
    q = p - (2.30753 + 0.27061 * p) / (1.0 + (0.99229 + 0.04481 * p) * p);
    if (1.0 < a && 1.0 < b) {
        const double r = (q * q - 3.0) / 6.0;
        const double s = 1.0 / (2.0 * a - 1.0);
        const double t = 1.0 / (2.0 * b - 1.0);
        const double h = 2.0 / (s + t);
        const double w = q * std::sqrt(h + r) / h - (t - s) * (r + 5.0 / 6.0 - 2.0 / (3.0 * h));
        p = a / (a + b * std::exp(2.0 * w));
    } else {
        double t = 1.0 / (9.0 * b);
        t = 2.0 * b * std::pow(1.0 - t + q * std::sqrt(t), 3.0);
        if (t <= 0.0) {
            p = 1.0 - std::exp((std::log((1.0 - x) * b) + lnb) / b);
        } else {
            t = 2.0 * (2.0 * a + b - 1.0) / t;
            if (t <= 1.0)
                p = std::exp((std::log(x * a) + lnb) / a);
            else
                p = 1.0 - 2.0 / (t + 1.0);
        }
    }
    if (p < 0.0001) p = 0.0001;
    if (p > 0.9999) p = 0.9999;
    const int e = (int)(-5.0 / a / a - 1.0 / std::pow(x, 0.2) - 13.0);
    const double acu = e > SAE ? std::pow(10.0, (double)e) : FPU;
    double pnext = 0.0, qprev = 0.0, sq = 1.0, prev = 1.0;
    for (;;) {
        bool err = false;
        q = beta_reg(a, b, p, &err);
        q = (q - x) * std::exp(lnb + (1.0 - a) * std::log(p) + (1.0 - b) * std::log(1.0 - p));
        if (!std::isfinite(q)) return NAN;  // the reference spins forever from here
        if (q * qprev <= 0.0) prev = sq > FPU ? sq : FPU;
        double g = 1.0;
        bool done = false;
        for (;;) {
            for (;;) {
                const double adj = g * q;
                sq = adj * adj;
                if (sq < prev) {
                    pnext = p - adj;
                    if (pnext >= 0. && pnext <= 1.) break;
                }
                g /= 3.0;
            }
            if (prev <= acu || q * q <= acu) {
                p = pnext;
                done = true;
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

// Acklam-style rational approximation refined by one Halley step: normal quantile to ~1e-16
double norm_ppf(double p) {
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                               1.383577518672690e+02,  -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                               6.680131188771972e+01, -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                               -2.549732539343734e+00, 4.374664141464968e+00,  2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
                               3.754408661907416e+00};
    double x;
    if (p < 0.02425) {
        const double q = std::sqrt(-2 * std::log(p));
        x = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
            ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    } else if (p <= 1 - 0.02425) {
        const double q = p - 0.5, r = q * q;
        x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
            (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1);
    } else {
        const double q = std::sqrt(-2 * std::log(1 - p));
        x = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
            ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    }
    for (int it = 0; it < 2; ++it) {  // Halley refinement on Phi(x) - p
        const double e = 0.5 * std::erfc(-x / std::sqrt(2.0)) - p;
        const double u = e * std::sqrt(2 * 3.14159265358979323846) * std::exp(x * x / 2);
        x = x - u / (1 + x * u / 2);
    }
    return x;
}

}  // namespace

double student_t_sf(double x, double df, bool* err) {
    if (err) *err = false;
    if (std::isinf(df)) return 0.5 * std::erfc(x / 1.41421356237309504880168872420969808);
    const double h = df / (df + x * x);
    const double ib = 0.5 * beta_reg(df / 2.0, 0.5, h, err);
    return x <= 0.0 ? 1.0 - ib : ib;
}

double student_t_ppf(double qv, double df) {
    const double x1 = qv >= 0.5 ? 1.0 - qv : qv;
    double y = inv_beta_reg(0.5 * df, 0.5, 2.0 * x1);
    double t;
    if (y == y) {
        t = std::sqrt(df * (1. - y) / y);
    } else {
        // reference does not terminate here (see header): Cornish-Fisher in 1/df around the normal quantile
        const double z = norm_ppf(1.0 - x1);
        const double z3 = z * z * z, z5 = z3 * z * z, z7 = z5 * z * z;
        t = z + (z3 + z) / (4 * df) + (5 * z5 + 16 * z3 + 3 * z) / (96 * df * df) +
            (3 * z7 + 19 * z5 + 17 * z3 - 15 * z) / (384 * df * df * df);
    }
    return qv >= 0.5 ? t : -t;
}

}  // namespace pds

extern "C" double pds_student_t_sf(double x, double df) {
    bool err = false;
    const double v = pds::student_t_sf(x, df, &err);
    return err ? NAN : v;
}
extern "C" double pds_student_t_ppf(double q, double df) { return pds::student_t_ppf(q, df); }
