/*
 * oracle/pds_perf.c -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg).
 *
 * The PERFORMANCE build of the CPU restatement: the same algorithm as pds_oracle.c (this file includes
 * it, one translation unit), compiled ON THE BOX THAT RUNS THE BENCH with
 *     gcc -O3 -march=native -fopenmp -fno-math-errno -fassociative-math -fno-signed-zeros -fno-trapping-math
 * (oracle/Makefile target `perf`), so FMA contraction, the widest vector unit of the host and vectorised
 * reductions are all allowed -- what faer's SIMD GEMM kernels do on the reference side.  The parity oracle
 * (libpds_oracle.so: -O2 -march=x86-64-v3 -ffp-contract=off, travels with the repo) stays the CHECKER; this
 * library is only ever timed, and bench.py reports its distance to the parity build on the sample.
 *
 * What is restated, with the reference's data movement (SURVEY.md 8(d) "CPU baseline beside it"):
 *   perf_grouped_lr_f64   group_by(key).agg(pds.lin_reg(...)): per group the marshalling copy into one column-major
 *                         buffer [y | X | 1] (series_to_mat_for_lr, src/num_ext/linear_regression.rs:164-185;
 *                         src/utils/mod.rs:118-132), X'X (+ lambda) and X'y (get_xtx_with_lambda / build_xty,
 *                         src/linear/lr/lr_solvers.rs:183-211, 262-278), then the gated column-pivoted QR
 *                         (faer_solve_lr_gated, :329-382 -- orc_gated_solve_gram of the included restatement).
 *                         OpenMP threads over contiguous group ranges stand in for Polars' rayon pool.
 *   perf_gram_cols_f64    the single regression's Gram build [X y]'[X y] straight from the column buffers.
 *   perf_stream_triad     a[i] = b[i] + s c[i] on all threads: what the host memory system delivers (context for the above).
 *   perf_alloc / perf_copy_first_touch   2 MiB-aligned storage whose pages are first written by the thread that later reads
 *                         them (static partition of the row axis), so a multi-socket host is not measured through one
 *                         NUMA node's memory controllers.
 */
#define _GNU_SOURCE
#include "pds_oracle.c"

#include <sys/mman.h>
#include <time.h>

static double perf_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void* perf_alloc(size_t bytes) {
    void* p = NULL;
    size_t al = (size_t)2 << 20;
    size_t sz = (bytes + al - 1) / al * al;
    if (sz == 0) sz = al;
    if (posix_memalign(&p, al, sz) != 0) return NULL;
#ifdef MADV_HUGEPAGE
    madvise(p, sz, MADV_HUGEPAGE);
#endif
    return p;
}

void perf_free(void* p) { free(p); }

/* thread t of nt owns rows [n t / nt, n (t+1) / nt): the same partition every timed routine below uses */
void perf_copy_first_touch(double* dst, const double* src, int64_t n, int nthreads) {
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        int tid = omp_get_thread_num(), nt = omp_get_num_threads();
        int64_t r0 = n * tid / nt, r1 = n * (tid + 1) / nt;
        memcpy(dst + r0, src + r0, sizeof(double) * (size_t)(r1 - r0));
    }
}

/* best-of-reps STREAM triad rate in GB/s (3 x 8 bytes per element, the STREAM convention) */
double perf_stream_triad(int64_t n, int nthreads, int reps) {
    if (nthreads < 1) nthreads = 1;
    double* a = (double*)perf_alloc(sizeof(double) * (size_t)n);
    double* b = (double*)perf_alloc(sizeof(double) * (size_t)n);
    double* c = (double*)perf_alloc(sizeof(double) * (size_t)n);
    if (!a || !b || !c) {
        free(a);
        free(b);
        free(c);
        return 0.0;
    }
#pragma omp parallel num_threads(nthreads)
    {
        int tid = omp_get_thread_num(), nt = omp_get_num_threads();
        int64_t r0 = n * tid / nt, r1 = n * (tid + 1) / nt;
        for (int64_t i = r0; i < r1; ++i) {
            a[i] = 0.0;
            b[i] = 1.0;
            c[i] = 2.0;
        }
    }
    double best = 0.0;
    for (int r = 0; r < reps + 1; ++r) {
        double t0 = perf_now();
#pragma omp parallel num_threads(nthreads)
        {
            int tid = omp_get_thread_num(), nt = omp_get_num_threads();
            int64_t r0 = n * tid / nt, r1 = n * (tid + 1) / nt;
            double* restrict pa = a;
            const double* restrict pb = b;
            const double* restrict pc = c;
#pragma omp simd
            for (int64_t i = r0; i < r1; ++i) pa[i] = pb[i] + 3.0 * pc[i];
        }
        double dt = perf_now() - t0;
        double gbps = 3.0 * 8.0 * (double)n / dt / 1e9;
        if (r > 0 && gbps > best) best = gbps; /* pass 0 warms the pages / threads */
    }
    volatile double sink = a[n / 2];
    (void)sink;
    free(a);
    free(b);
    free(c);
    return best;
}

/* upper triangle of Z'Z of the rows [r0, r1) of q columns, added into acc (q x q column-major) */
static inline void perf_gram_block(const double* const* cols, int q, int64_t r0, int64_t r1, double* acc) {
    for (int j = 0; j < q; ++j) {
        const double* restrict cj = cols[j];
        for (int i = 0; i <= j; ++i) {
            const double* restrict ci = cols[i];
            double s = 0.0;
#pragma omp simd reduction(+ : s)
            for (int64_t r = r0; r < r1; ++r) s += ci[r] * cj[r];
            acc[i + (size_t)j * q] += s;
        }
    }
}

#define PERF_GRAM_BLOCK 512 /* rows per cache block: 17 columns x 512 x 8 B = 68 KB, L2 resident */

/* [cols]'[cols], q columns of n rows, out q x q column-major; returns the seconds of the best of `reps` passes */
double perf_gram_cols_f64(const double* const* cols, int64_t n, int q, double* out, int nthreads, int reps) {
    if (nthreads < 1) nthreads = 1;
    double* tot = (double*)calloc((size_t)nthreads * q * q, sizeof(double));
    double best = 1e300;
    for (int rep = 0; rep < reps + 1; ++rep) {
        memset(tot, 0, sizeof(double) * (size_t)nthreads * q * q);
        double t0 = perf_now();
#pragma omp parallel num_threads(nthreads)
        {
            int tid = omp_get_thread_num(), nt = omp_get_num_threads();
            int64_t a0 = n * tid / nt, a1 = n * (tid + 1) / nt;
            double* mine = tot + (size_t)tid * q * q;
            for (int64_t r0 = a0; r0 < a1; r0 += PERF_GRAM_BLOCK) {
                int64_t r1 = r0 + PERF_GRAM_BLOCK < a1 ? r0 + PERF_GRAM_BLOCK : a1;
                perf_gram_block(cols, q, r0, r1, mine);
            }
        }
        double dt = perf_now() - t0;
        if (rep > 0 && dt < best) best = dt;
    }
    for (int j = 0; j < q; ++j)
        for (int i = 0; i <= j; ++i) {
            double s = 0.0;
            for (int t = 0; t < nthreads; ++t) s += tot[(size_t)t * q * q + i + (size_t)j * q];
            out[i + (size_t)j * q] = s;
            out[j + (size_t)i * q] = s;
        }
    free(tot);
    return best;
}

/*
 * cols = [y, x1..xp]; group g = rows [off[g], off[g+1]).  coeffs n_groups x (p + bias) row-major; flags[g] = 1 for a
 * null group.  Thread t owns the contiguous group range [G t / nt, G (t+1) / nt).  One warm-up pass over the first
 * groups of every thread's range, then `passes` timed passes over all groups; returns the seconds of ALL timed passes.
 */
double perf_grouped_lr_f64(const double* const* cols, int p, const int64_t* off, int64_t n_groups, int add_bias,
                           double lambda, double tol, double* coeffs, unsigned char* flags, int nthreads, int passes) {
    const int pp = p + (add_bias ? 1 : 0);
    if (nthreads < 1) nthreads = 1;
    int64_t max_m = 0;
    for (int64_t g = 0; g < n_groups; ++g)
        if (off[g + 1] - off[g] > max_m) max_m = off[g + 1] - off[g];
    double elapsed = 0.0;
    for (int pass = -1; pass < passes; ++pass) {
        double t0 = perf_now();
#pragma omp parallel num_threads(nthreads)
        {
            int tid = omp_get_thread_num(), nt = omp_get_num_threads();
            int64_t g0 = n_groups * tid / nt, g1 = n_groups * (tid + 1) / nt;
            if (pass < 0 && g1 > g0 + 64) g1 = g0 + 64; /* warm-up: thread start, scratch pages, code */
            double* buf = (double*)malloc(sizeof(double) * (size_t)(max_m > 0 ? max_m : 1) * (size_t)(pp + 1));
            double* gm = (double*)malloc(sizeof(double) * (size_t)pp * pp);
            const double* xc[ORC_MAX_P + 1];
            for (int64_t g = g0; g < g1; ++g) {
                const int64_t r0 = off[g], m = off[g + 1] - off[g];
                double* out = coeffs + (size_t)g * pp;
                if (m < pp || m == 0) {
                    for (int j = 0; j < pp; ++j) out[j] = NAN;
                    flags[g] = 1;
                    continue;
                }
                /* the marshalling copy: [y | X | 1] column-major */
                memcpy(buf, cols[0] + r0, sizeof(double) * (size_t)m);
                for (int j = 0; j < p; ++j) memcpy(buf + (size_t)(j + 1) * m, cols[j + 1] + r0, sizeof(double) * (size_t)m);
                if (add_bias)
                    for (int64_t r = 0; r < m; ++r) buf[(size_t)(p + 1) * m + r] = 1.0;
                for (int j = 0; j < pp; ++j) xc[j] = buf + (size_t)(j + 1) * m;
                memset(gm, 0, sizeof(double) * (size_t)pp * pp);
                perf_gram_block(xc, pp, 0, m, gm);
                for (int j = 0; j < pp; ++j)
                    for (int i = 0; i < j; ++i) gm[j + (size_t)i * pp] = gm[i + (size_t)j * pp];
                if (lambda > 0)
                    for (int j = 0; j < pp - (add_bias ? 1 : 0); ++j) gm[j + (size_t)j * pp] += lambda;
                for (int j = 0; j < pp; ++j) {
                    const double* restrict cj = xc[j];
                    const double* restrict yy = buf;
                    double s = 0.0;
#pragma omp simd reduction(+ : s)
                    for (int64_t r = 0; r < m; ++r) s += cj[r] * yy[r];
                    out[j] = s;
                }
                int ok = tol > 0 ? orc_gated_solve_gram_f64(gm, pp, out, 1, 0, tol) : (orc_solve_xtx_xty_f64(gm, pp, out, 1, 0), 1);
                if (!ok)
                    for (int j = 0; j < pp; ++j) out[j] = NAN;
                flags[g] = ok ? 0 : 1;
            }
            free(buf);
            free(gm);
        }
        if (pass >= 0) elapsed += perf_now() - t0;
    }
    return elapsed;
}

int perf_num_procs(void) { return omp_get_num_procs(); }
