/*
 * oracle/pds_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C declarations of the CPU restatement of polars_ds's least-squares path (see pds_oracle.c for
 * the provenance / parity statement).  The product library (include/pds_lstsq.h) does not depend on
 * this header or on libpds_oracle.so.
 */
#ifndef PDS_ORACLE_H
#define PDS_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* special functions -- ref: src/stats_utils/{gamma,beta}.rs */
double orc_ln_gamma(double x);
double orc_beta_reg(double a, double b, double x, int* err);
double orc_inv_beta_reg(double a, double b, double x);
double orc_student_t_sf(double x, double df, int* err);
double orc_student_t_ppf(double q, double df);
int orc_max_threads(void);

#define ORC_DECLARE(REAL, S)                                                                       \
    typedef struct {                                                                               \
        int n;                                                                                     \
        REAL* a;                                                                                   \
        REAL* tau;                                                                                 \
        int* jpvt;                                                                                 \
    } orc_qr##S;                                                                                   \
    void orc_gram_cols##S(const REAL* const* cols, int64_t n, int p, REAL* out, int nthreads);    \
    void orc_gram##S(const REAL* x, int64_t n, int p, REAL* out, int nthreads);                   \
    void orc_xty##S(const REAL* x, const REAL* y, int64_t n, int p, int k, REAL* out,             \
                    int nthreads);                                                                 \
    void orc_qr_factor##S(const REAL* g, int n, orc_qr##S* q);                                    \
    void orc_qr_solve##S(const orc_qr##S* q, REAL* b, int nrhs);                                  \
    void orc_qr_inverse##S(const orc_qr##S* q, REAL* inv);                                        \
    REAL orc_qr_sum_ln_absdiag##S(const orc_qr##S* q);                                            \
    void orc_qr_free##S(orc_qr##S* q);                                                            \
    int orc_llt_factor##S(const REAL* g, int n, REAL* l);                                         \
    void orc_llt_solve##S(const REAL* l, int n, REAL* b, int nrhs);                               \
    int orc_svd##S(const REAL* a, int n, REAL* u, REAL* s, REAL* v);                              \
    void orc_solve_xtx_xty##S(const REAL* g, int p, REAL* b, int k, int solver);                  \
    void orc_solve_lr##S(const REAL* x, const REAL* y, int64_t n, int p, int k, REAL lambda,      \
                         int add_bias, int solver, REAL* beta, int nthreads);                     \
    int orc_gated_solve_gram##S(const REAL* g, int p, REAL* xty, int k, int solver, REAL tol);    \
    int orc_solve_lr_gated##S(const REAL* x, const REAL* y, int64_t n, int p, int k, REAL lambda, \
                              int add_bias, int solver, REAL tol, REAL* beta, int nthreads);      \
    int orc_solve_lr_rcond##S(const REAL* x, const REAL* y, int64_t n, int p, REAL lambda,        \
                              int add_bias, REAL rcond, REAL* beta, REAL* singular_values,        \
                              int nthreads);                                                       \
    void orc_grouped_lr##S(const REAL* const* cols, int p, const int64_t* off, int64_t n_groups,  \
                           int add_bias, REAL lambda, int solver, REAL tol, REAL* coeffs,         \
                           unsigned char* flags, int nthreads);                                    \
    void orc_weighted_lr##S(const REAL* x, const REAL* y, const REAL* w, int64_t n, int p,        \
                            int solver, REAL* beta);                                               \
    int orc_glm_irls##S(const REAL* x, const REAL* y, int64_t n, int p, int link, int variance,   \
                        REAL tol, int max_iter, REAL* beta);                                       \
    int orc_cd_from_gram##S(const REAL* g, const REAL* xty, const REAL* col_sums, REAL y_sum,     \
                            REAL m, int p, REAL l1_reg, REAL l2_reg, int add_bias, REAL tol,      \
                            int max_iter, int positive, REAL* beta, int* converged);              \
    int orc_coordinate_descent##S(const REAL* x, const REAL* y, int64_t n, int p, REAL l1_reg,    \
                                  REAL l2_reg, int add_bias, REAL tol, int max_iter,              \
                                  int positive, REAL* beta, int* converged, int nthreads);        \
    int orc_nnls_from_gram##S(const REAL* g, const REAL* xty, int p, int add_bias, REAL tol,      \
                              int max_iter, REAL* beta);                                           \
    int orc_nn_lr##S(const REAL* x, const REAL* y, int64_t n, int p, int add_bias, REAL tol,      \
                     int max_iter, REAL* beta, int nthreads);                                      \
    void orc_qr_lr_with_inv##S(const REAL* x, int64_t ldx, const REAL* y, int64_t n, int p,       \
                               REAL lambda, int add_bias, REAL* inv, REAL* beta);                 \
    void orc_woodbury_step##S(REAL* inv, REAL* beta, int p, const REAL* xrow, int64_t ldx,        \
                              REAL y, REAL c);                                                     \
    void orc_recursive_lr##S(const REAL* x, const REAL* y, int64_t n, int p, int64_t n0,          \
                             REAL lambda, REAL* coeffs);                                           \
    void orc_rolling_lr##S(const REAL* x, const REAL* y, int64_t n, int p, int64_t win,           \
                           REAL lambda, REAL* coeffs);                                             \
    int64_t orc_rolling_skipping_lr##S(const REAL* x, const REAL* y, int64_t n, int p,            \
                                       int64_t win, int64_t min_size, REAL lambda, REAL* coeffs,  \
                                       unsigned char* valid);                                      \
    void orc_lin_reg_report##S(const REAL* x, const REAL* y, int64_t n, int p, REAL y_var,        \
                               int se_type, REAL* beta, REAL* std_err, REAL* tval, REAL* pval,    \
                               REAL* ci_lo, REAL* ci_hi, REAL* r2_out, REAL* adj_r2_out);         \
    void orc_wls_report##S(const REAL* x, const REAL* y, const REAL* w, int64_t n, int p,         \
                           REAL y_var, REAL* beta, REAL* std_err, REAL* tval, REAL* pval,         \
                           REAL* ci_lo, REAL* ci_hi, REAL* r2_out, REAL* adj_r2_out);

ORC_DECLARE(double, _f64)
ORC_DECLARE(float, _f32)

#ifdef __cplusplus
}
#endif
#endif
