// solve.hip -- the small p' x p' solves, batched, LDS resident.
//
// Replaces, for one or a million systems at a time:
//   solve_xtx_xty / faer_solve_lr / faer_solve_lr_gated   lr_solvers.rs:282-382
//   faer_qr_lr_with_inv (the (X'X)^-1 by col-piv QR)       lr_online_solvers.rs:120-143
//   faer_coordinate_descent sweeps                          lr_solvers.rs:477-537
//   faer_nn_lr sweeps                                       lr_solvers.rs:569-599
//
// Layout: a sub-wave group of LPS lanes (LPS = 4..64, power of two >= p') owns one system; 64/LPS
// systems per wave.  Lane j of the group owns column j of G = X'X (+lambda) in LDS (odd leading
// dimension => conflict-free column-parallel access).  The factorisation is Householder QR with column
// pivoting on G itself -- the reference's default `col_piv_qr` on the normal equations -- so the rank
// gate statistic sum(ln|R_ii|) - sum(ln G_ii) is the reference's (lr_solvers.rs:341-357).
// All arithmetic is f64 for both precisions (the f32 twin of the reference solves in f32; solving the
// same f32-accumulated moments in f64 is strictly closer to the exact answer).
#include "common.hpp"

namespace pds {

#define WSYNC() PDS_WAVE_LDS_SYNC()

struct SolveDev {
    int p, pp, bias, solver, want_inv, lambda_on_bias;
    double lambda, ln_tol;  // ln_tol = ln(gate_tol); gate disabled when gate_on == 0
    int gate_on;
};

template <int LPS>
__device__ __forceinline__ double grp_sum(double v) {
#pragma unroll
    for (int o = LPS / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, LPS);
    return v;
}

template <int LPS>
__device__ __forceinline__ void grp_argmax(double& v, int& idx) {
#pragma unroll
    for (int o = LPS / 2; o >= 1; o >>= 1) {
        double ov = __shfl_xor(v, o, LPS);
        int oi = __shfl_xor(idx, o, LPS);
        if (ov > v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
}

// doubles of LDS one system needs
__host__ __device__ inline int sys_doubles(int pp, int want_inv) {
    const int ld = pp | 1;
    const int nrhs = 1 + (want_inv ? pp : 0);
    return ld * pp + ld * nrhs + 2 * pp + ((pp + 1) / 2) + 2;  // A, B, vn1, vn2, perm(int), pad
}

template <typename T, int LPS>
__global__ __launch_bounds__(256) void solve_kernel(const T* __restrict__ moments, int64_t n_sys, SolveDev sp,
                                                    T* __restrict__ coeffs, uint8_t* __restrict__ flags,
                                                    T* __restrict__ inv_out,
                                                    const int64_t* __restrict__ rows_per_sys) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int SPW = 64 / LPS;  // systems per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPS, j = lane % LPS;
    const int p = sp.p, pp = sp.pp, q = p + 2, ld = pp | 1;
    const int nrhs = 1 + (sp.want_inv ? pp : 0);
    const int sd = sys_doubles(pp, sp.want_inv);
    double* A = sm + (size_t)((wave * SPW + sub) * sd);
    double* B = A + ld * pp;
    double* vn1 = B + ld * nrhs;
    double* vn2 = vn1 + pp;
    int* perm = reinterpret_cast<int*>(vn2 + pp);

    const int64_t sys_per_block = (int64_t)(blockDim.x >> 6) * SPW;
    for (int64_t base = (int64_t)blockIdx.x * sys_per_block; base < n_sys; base += (int64_t)gridDim.x * sys_per_block) {
        const int64_t sys = base + wave * SPW + sub;
        const bool live = sys < n_sys;  // whole sub-group uniform
        const T* M = moments + (live ? sys : 0) * (int64_t)(q * q);
        // ---- load G and rhs
        for (int idx = j; idx < pp * pp; idx += LPS) {
            const int r = idx % pp, c = idx / pp;
            double v = (double)M[r + c * q];
            if (r == c && sp.lambda > 0.0 && (r < p || sp.lambda_on_bias)) v += sp.lambda;
            A[r + c * ld] = v;
        }
        for (int r = j; r < pp; r += LPS) B[r] = (double)M[r + (p + 1) * q];
        if (sp.want_inv)
            for (int idx = j; idx < pp * pp; idx += LPS) {
                const int r = idx % pp, c = idx / pp;
                B[r + (c + 1) * ld] = (r == c) ? 1.0 : 0.0;
            }
        WSYNC();
        bool is_null = false;
        if (rows_per_sys) {  // per-group `pl_lr` errors with "#Data < #features" -> reported as null
            if (live && rows_per_sys[sys + 1] - rows_per_sys[sys] < pp) is_null = true;
        }
        // ---- gate denominator: sum ln diag, non-positive diagonal => gated (lr_solvers.rs:341-347)
        double ln_den = 0.0;
        if (sp.gate_on) {
            double dj = (j < pp) ? A[j + j * ld] : 1.0;
            int bad = (dj <= 0.0) ? 1 : 0;
#pragma unroll
            for (int o = LPS / 2; o >= 1; o >>= 1) bad |= __shfl_xor(bad, o, LPS);
            if (bad) is_null = true;
            ln_den = grp_sum<LPS>((j < pp) ? log(dj) : 0.0);
        }
        double ln_det = 0.0;
        bool chol_done = false;
        if (sp.solver == PDS_SOLVER_CHOLESKEY) {
            // ---- Cholesky (llt(Side::Lower)) on a copy-free basis: A is overwritten by L (lower)
            bool ok = true;
            for (int k = 0; k < pp; ++k) {
                double d = A[k + k * ld];
                if (!(d > 0.0) || !isfinite(d)) {
                    ok = false;
                    break;
                }
                const double lkk = sqrt(d);
                WSYNC();
                if (j == k) A[k + k * ld] = lkk;
                if (j > k && j < pp) A[j + k * ld] = A[j + k * ld] / lkk;  // column k below the diagonal
                WSYNC();
                // trailing update: lane j owns column j (j > k): A[i,j] -= L[i,k] L[j,k] for i >= j
                if (j > k && j < pp) {
                    const double ljk = A[j + k * ld];
                    for (int i = j; i < pp; ++i) A[i + j * ld] -= A[i + k * ld] * ljk;
                }
                WSYNC();
            }
            if (ok) {
                double s = grp_sum<LPS>((j < pp) ? log(A[j + j * ld]) : 0.0);
                ln_det = s + s;
                // forward / backward substitution, lane c owns rhs column c
                for (int c = j; c < nrhs; c += LPS) {
                    double* b = B + c * ld;
                    for (int i = 0; i < pp; ++i) {
                        double s2 = b[i];
                        for (int k = 0; k < i; ++k) s2 -= A[i + k * ld] * b[k];
                        b[i] = s2 / A[i + i * ld];
                    }
                    for (int i = pp - 1; i >= 0; --i) {
                        double s2 = b[i];
                        for (int k = i + 1; k < pp; ++k) s2 -= A[k + i * ld] * b[k];
                        b[i] = s2 / A[i + i * ld];
                    }
                }
                for (int r = j; r < pp; r += LPS) perm[r] = r;
                WSYNC();
                chol_done = true;
            } else if (sp.gate_on) {
                is_null = true;  // "Not positive-definite -> rank-deficient" lr_solvers.rs:370-371
                chol_done = true;
            } else {
                // ungated failure falls back to QR (lr_solvers.rs:288-291): reload G
                WSYNC();
                for (int idx = j; idx < pp * pp; idx += LPS) {
                    const int r = idx % pp, c = idx / pp;
                    double v = (double)M[r + c * q];
                    if (r == c && sp.lambda > 0.0 && (r < p || sp.lambda_on_bias)) v += sp.lambda;
                    A[r + c * ld] = v;
                }
                WSYNC();
            }
        }
        if (!chol_done) {
            // ---- Householder QR with column pivoting (xGEQP2 scheme), column j <-> lane j
            if (j < pp) {
                double s = 0.0;
                for (int i = 0; i < pp; ++i) s = fma(A[i + j * ld], A[i + j * ld], s);
                vn1[j] = vn2[j] = sqrt(s);
                perm[j] = j;
            }
            WSYNC();
            const double tol3z = 1.4901161193847656e-08;  // sqrt(eps_f64)
            for (int k = 0; k < pp; ++k) {
                double best = (j >= k && j < pp) ? vn1[j] : -1.0;
                if (best != best) best = -1.0;  // NaN never wins a pivot
                int pv = j;
                grp_argmax<LPS>(best, pv);
                if (best < 0.0) pv = k;
                if (pv != k) {  // swap columns k <-> pv (lane i swaps row i) and their bookkeeping
                    for (int i = j; i < pp; i += LPS) {
                        const double t = A[i + pv * ld];
                        A[i + pv * ld] = A[i + k * ld];
                        A[i + k * ld] = t;
                    }
                    if (j == 0) {
                        const int tp = perm[pv];
                        perm[pv] = perm[k];
                        perm[k] = tp;
                        vn1[pv] = vn1[k];
                        vn2[pv] = vn2[k];
                    }
                }
                WSYNC();
                // reflector from column k
                double part = 0.0;
                for (int i = k + 1 + j; i < pp; i += LPS) part = fma(A[i + k * ld], A[i + k * ld], part);
                const double xn2 = grp_sum<LPS>(part);
                const double alpha = A[k + k * ld];
                double tau = 0.0;
                WSYNC();
                if (xn2 != 0.0) {
                    double beta = sqrt(alpha * alpha + xn2);
                    if (alpha > 0.0) beta = -beta;
                    tau = (beta - alpha) / beta;
                    const double sc = 1.0 / (alpha - beta);
                    for (int i = k + 1 + j; i < pp; i += LPS) A[i + k * ld] *= sc;
                    if (j == 0) A[k + k * ld] = beta;
                }
                WSYNC();
                if (tau != 0.0) {
                    // trailing columns of A and every rhs column: lane handles columns c = j, j+LPS, ...
                    const int ntot = pp + nrhs;
                    for (int c = j; c < ntot; c += LPS) {
                        if (c <= k) continue;
                        double* col = (c < pp) ? (A + c * ld) : (B + (c - pp) * ld);
                        double w = col[k];
                        for (int i = k + 1; i < pp; ++i) w = fma(A[i + k * ld], col[i], w);
                        w *= tau;
                        col[k] -= w;
                        for (int i = k + 1; i < pp; ++i) col[i] = fma(-w, A[i + k * ld], col[i]);
                    }
                }
                WSYNC();
                // partial column norm downdate (same safeguard as the oracle / LAPACK)
                if (j > k && j < pp && vn1[j] != 0.0) {
                    double t = fabs(A[k + j * ld]) / vn1[j];
                    t = 1.0 - t * t;
                    if (t < 0.0) t = 0.0;
                    const double r = vn1[j] / vn2[j];
                    const double t2 = t * r * r;
                    if (t2 <= tol3z) {
                        double s = 0.0;
                        for (int i = k + 1; i < pp; ++i) s = fma(A[i + j * ld], A[i + j * ld], s);
                        vn1[j] = vn2[j] = sqrt(s);
                    } else
                        vn1[j] *= sqrt(t);
                }
                WSYNC();
            }
            ln_det = grp_sum<LPS>((j < pp) ? log(fabs(A[j + j * ld])) : 0.0);
            // back substitution.  rhs 0 (the hot one): row-parallel with a group reduction per row;
            // identity columns (inverse): lane-per-column serial.
            {
                double zj = 0.0;  // lane j ends up holding z_j
                for (int i = pp - 1; i >= 0; --i) {
                    const double t = (j > i && j < pp) ? A[i + j * ld] * zj : 0.0;
                    const double s = grp_sum<LPS>(t);
                    const double zi = (B[i] - s) / A[i + i * ld];
                    if (j == i) zj = zi;
                }
                WSYNC();
                if (j < pp) B[j] = zj;
            }
            for (int c = 1 + j; c < nrhs; c += LPS) {
                double* b = B + c * ld;
                for (int i = pp - 1; i >= 0; --i) {
                    double s = b[i];
                    for (int m = i + 1; m < pp; ++m) s -= A[i + m * ld] * b[m];
                    b[i] = s / A[i + i * ld];
                }
            }
            WSYNC();
        }
        if (sp.gate_on && !is_null) {
            if (ln_det - ln_den <= sp.ln_tol) is_null = true;  // NaN compares false, like the reference
        }
        // ---- write back (un-permute): x[perm[k]] = z[k]
        if (live) {
            T* out = coeffs + sys * (int64_t)pp;
            const double nanv = __builtin_nan("");
            if (j < pp) out[perm[j]] = is_null ? (T)nanv : (T)B[j];
            if (j == 0 && flags) flags[sys] = is_null ? 1 : 0;
            if (sp.want_inv && inv_out) {
                T* io = inv_out + sys * (int64_t)(pp * pp);
                for (int idx = j; idx < pp * pp; idx += LPS) {
                    const int r = idx % pp, c = idx / pp;
                    io[perm[r] + c * pp] = (T)B[r + (c + 1) * ld];
                }
            }
        }
        WSYNC();
    }
}

template <typename T, int LPS>
static int launch_solve_lps(pds_ctx* ctx, const T* d_moments, int64_t n_sys, const SolveDev& sd, T* d_coeffs,
                            uint8_t* d_flags, T* d_inv_out, const int64_t* d_rows) {
    constexpr int SPW = 64 / LPS;
    const int per_sys = sys_doubles(sd.pp, sd.want_inv);
    int waves = 4;
    while (waves > 1 && (size_t)waves * SPW * per_sys * 8 > 60 * 1024) waves >>= 1;
    const size_t lds = (size_t)waves * SPW * per_sys * 8;
    if (lds > 160 * 1024) return fail(PDS_ERR_UNSUPPORTED, "solve: system too large for LDS");
    const int64_t spb = (int64_t)waves * SPW;
    int64_t nb = (n_sys + spb - 1) / spb;
    nb = std::min<int64_t>(nb, (int64_t)ctx->num_cus * 8);
    if (lds > 64 * 1024)
        PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<T, LPS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    KernelTimer timer(ctx, kKindSolve);
    hipLaunchKernelGGL((solve_kernel<T, LPS>), dim3((unsigned)nb), dim3(waves * 64), lds, ctx->stream, d_moments,
                       n_sys, sd, d_coeffs, d_flags, d_inv_out, d_rows);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

template <typename T>
int launch_solve(pds_ctx* ctx, const T* d_moments, int64_t n_sys, const SolveParams& sp, T* d_coeffs,
                 uint8_t* d_flags, T* d_inv_out, const int64_t* d_rows_per_sys) {
    SolveDev sd;
    sd.p = sp.p;
    sd.bias = sp.add_bias ? 1 : 0;
    sd.pp = sp.p + sd.bias;
    sd.solver = sp.solver;
    sd.want_inv = d_inv_out ? 1 : 0;
    sd.lambda_on_bias = sp.lambda_on_bias;
    sd.lambda = sp.lambda;
    sd.gate_on = sp.gate_tol > 0.0 ? 1 : 0;
    sd.ln_tol = sd.gate_on ? std::log(sp.gate_tol) : 0.0;
    if (n_sys <= 0) return PDS_OK;
    const int pp = sd.pp;
    // hot path: register-resident pivoted QR (solve_reg.hip) for p' <= 16 when no inverse is wanted
    if (pp >= 1 && pp <= 16 && !d_inv_out && (sp.solver != PDS_SOLVER_CHOLESKEY || sp.gate_tol > 0.0))
        return launch_solve_reg<T>(ctx, d_moments, n_sys, sp, d_coeffs, d_flags, d_rows_per_sys);
    if (pp < 1) return fail(PDS_ERR_INVALID, "solve: no coefficients");
    if (pp > 64) {
        if (d_rows_per_sys) return fail(PDS_ERR_UNSUPPORTED, "grouped solve: at most 64 coefficients");
        return launch_solve_big<T>(ctx, d_moments, n_sys, sp, d_coeffs, d_flags, d_inv_out);
    }
    if (pp <= 4) return launch_solve_lps<T, 4>(ctx, d_moments, n_sys, sd, d_coeffs, d_flags, d_inv_out, d_rows_per_sys);
    if (pp <= 8) return launch_solve_lps<T, 8>(ctx, d_moments, n_sys, sd, d_coeffs, d_flags, d_inv_out, d_rows_per_sys);
    if (pp <= 16) return launch_solve_lps<T, 16>(ctx, d_moments, n_sys, sd, d_coeffs, d_flags, d_inv_out, d_rows_per_sys);
    if (pp <= 32) return launch_solve_lps<T, 32>(ctx, d_moments, n_sys, sd, d_coeffs, d_flags, d_inv_out, d_rows_per_sys);
    return launch_solve_lps<T, 64>(ctx, d_moments, n_sys, sd, d_coeffs, d_flags, d_inv_out, d_rows_per_sys);
}

// =============================================================================================
// coordinate descent / NNLS on the moment matrix: one wavefront, beta and the scalars in LDS,
// Gram columns read from L2 (the matrix is tiny next to the 4 MiB L2: 1 MiB at p = 512 f32)
// =============================================================================================
// Batched CD / NNLS (one workgroup per group): a group with fewer rows than coefficients is what per-group `pl_lr`
// rejects with "#Data < #features" (linear_regression.rs:169-173) -> null, like solve_kernel does for OLS.
template <typename T>
__device__ __forceinline__ bool batched_too_few_rows(const int64_t* __restrict__ rows_per_sys, int pp, T* coeffs,
                                                     uint8_t* flags) {
    bool small = false;
    if (rows_per_sys) small = rows_per_sys[blockIdx.x + 1] - rows_per_sys[blockIdx.x] < pp;
    if (small)
        for (int i = threadIdx.x; i < pp; i += 64) coeffs[i] = (T)NAN;
    if (flags && threadIdx.x == 0) flags[blockIdx.x] = small ? 1 : 0;
    return small;
}

// The reference (lr_solvers.rs:486-530) recomputes dot_j = sum_{k != j} G[k,j] beta_k for every coordinate of every
// sweep: p^2 work per sweep and a dependent L2 round trip per coordinate.  Here the same iteration is carried by the
// gradient  r_j = (X'y)_j - sum_k G[k,j] beta_k  (all k), kept in LDS: the reference's
//   main_update_j = (X'y)_j - dot_j  is  r_j + G[j,j] beta_j,
// and r only moves when a coefficient moves (r -= delta * G[:,j], one coalesced column read).  Coordinates are
// evaluated 64 at a time against the current r; the first lane whose coefficient would change commits it, the
// evaluation restarts behind it -- lanes in front of it saw exactly the state the sequential sweep would have
// shown them, so the iterates are the reference's up to f64 rounding of the running r.  A sweep costs
// (p / 64 + #coefficients that move) steps instead of p.
template <typename T>
__global__ __launch_bounds__(64) void cd_kernel(const T* __restrict__ M, int p, int bias, double l1_reg,
                                                double l2_reg, double tol, int max_iter, int positive,
                                                T* __restrict__ coeffs, int* __restrict__ info,
                                                uint8_t* __restrict__ flags, const int64_t* __restrict__ rows_per_sys) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x;
    const int pp = p + bias, q = p + 2;
    // batched form (grouped regressions): workgroup b owns system b
    M += (int64_t)blockIdx.x * q * q;
    coeffs += (int64_t)blockIdx.x * pp;
    if (batched_too_few_rows(rows_per_sys, pp, coeffs, flags)) return;
    double* beta = sm;           // pp
    double* r = sm + (p + 2);    // pp
    double* gd = r + (p + 2);    // p: G[j,j]
    const double m = (double)M[p + p * q];  // n (row count) lives in the bias/bias slot
    const double lambda_l1 = m * l1_reg, ridge = m * l2_reg;
    for (int i = lane; i < pp; i += 64) {
        beta[i] = 0.0;
        r[i] = (double)M[i + (p + 1) * q];  // X'y, and sum y in the bias slot
        if (i < p) gd[i] = (double)M[i + i * q];
    }
    WSYNC();
    constexpr int kCdPrefetch = 9;  // 64 x 9 >= 514 = the moment columns of config 5; wider systems read each column when it is due
    T pf[kCdPrefetch];
    int pf_col = -1;
#pragma unroll
    for (int u = 0; u < kCdPrefetch; ++u) pf[u] = T(0);
    int it = 0, conv = 0;
    for (it = 0; it < max_iter; ++it) {
        double max_change = 0.0;
        int j0 = 0;
        while (j0 < p) {
            const int j = j0 + lane;
            double before = 0.0, after = 0.0;
            if (j < p) {
                before = beta[j];
                const double g = gd[j];
                const double main_update = fma(g, before, r[j]);
                if (positive && main_update < 0.0)
                    after = 0.0;
                else {
                    const double sg = (main_update < 0.0 || (main_update == 0.0 && signbit(main_update))) ? -1.0 : 1.0;
                    const double mag = fabs(main_update) - lambda_l1;
                    after = sg * (mag > 0.0 ? mag : 0.0) / (g + ridge);
                }
            }
            const unsigned long long moved = __ballot(j < p && after != before);
            if (moved == 0ull) {
                j0 += 64;
                continue;
            }
            const int f = __ffsll((long long)moved) - 1;
            const int jc = j0 + f;
            const double delta = __shfl(after - before, f);
            const double committed = __shfl(after, f);
            WSYNC();
            if (lane == 0) beta[jc] = committed;
            const T* col = M + (int64_t)jc * q;
            // The column of a moving coordinate comes from L2 / HBM (~1 us) and the next coordinate cannot start before the
            // residuals are updated: a sweep over a 512-feature elastic net costs (moving coordinates) x (that latency).  The
            // next candidate is known now -- the next set bit of `moved`, unless this update changes its mind -- so its column is
            // fetched into registers while this one is applied.
            if (pp <= 64 * kCdPrefetch) {
                T cur[kCdPrefetch];
                const bool hit = pf_col == jc;
#pragma unroll
                for (int u = 0; u < kCdPrefetch; ++u) {
                    const int k = lane + 64 * u;
                    cur[u] = hit ? pf[u] : (k < pp ? col[k] : T(0));
                }
                const unsigned long long rest = moved & ~((2ull << f) - 1ull);  // candidates behind jc in this chunk
                pf_col = rest ? j0 + (__ffsll((long long)rest) - 1) : -1;
                if (pf_col >= 0) {
                    const T* nxt = M + (int64_t)pf_col * q;
#pragma unroll
                    for (int u = 0; u < kCdPrefetch; ++u) {
                        const int k = lane + 64 * u;
                        pf[u] = k < pp ? nxt[k] : T(0);
                    }
                }
#pragma unroll
                for (int u = 0; u < kCdPrefetch; ++u) {
                    const int k = lane + 64 * u;
                    if (k < pp) r[k] = fma(-delta, (double)cur[u], r[k]);
                }
            } else {
                for (int k = lane; k < pp; k += 64) r[k] = fma(-delta, (double)col[k], r[k]);
            }
            WSYNC();
            const double d = fabs(delta);
            max_change = d > max_change ? d : max_change;
            j0 = jc + 1;
        }
        if (bias) {  // bias = (sum y - sum_j beta_j colsum_j) / m = (r_p + m beta_p) / m   (:514-522)
            const double old = beta[p];
            const double nb = fma(m, old, r[p]) / m;
            const double delta = nb - old;
            WSYNC();
            if (lane == 0) beta[p] = nb;
            const T* col = M + (int64_t)p * q;
            for (int k = lane; k < pp; k += 64) r[k] = fma(-delta, (double)col[k], r[k]);
            WSYNC();
        }
        conv = max_change < tol;
        if (conv) {
            ++it;
            break;
        }
    }
    for (int i = lane; i < pp; i += 64) coeffs[i] = (T)beta[i];
    if (lane == 0 && info) {
        info[2 * blockIdx.x] = it;
        info[2 * blockIdx.x + 1] = conv;
    }
}

template <typename T>
__global__ __launch_bounds__(64) void nnls_kernel(const T* __restrict__ M, int p, int bias, double tol,
                                                  int max_iter, T* __restrict__ coeffs, uint8_t* __restrict__ flags,
                                                  const int64_t* __restrict__ rows_per_sys) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x;
    const int pp = p + bias, q = p + 2;
    M += (int64_t)blockIdx.x * q * q;
    coeffs += (int64_t)blockIdx.x * pp;
    if (batched_too_few_rows(rows_per_sys, pp, coeffs, flags)) return;
    double* beta = sm;
    double* mu = sm + pp;
    for (int i = lane; i < pp; i += 64) {
        beta[i] = 0.0;
        mu[i] = -(double)M[i + (p + 1) * q];
    }
    WSYNC();
    for (int it = 0; it < max_iter; ++it) {
        int ok = 1;
        for (int i = lane; i < pp; i += 64) {
            if (!(mu[i] >= -tol)) ok = 0;
            if (beta[i] > 0.0 && !(mu[i] <= tol)) ok = 0;
        }
        ok = __all(ok);
        if (ok) break;
        for (int k = 0; k < pp; ++k) {
            const double beta_k = beta[k];
            double update = beta_k - mu[k] / (double)M[k + k * q];
            if (!bias || k < pp - 1) update = (update > 0.0) ? update : 0.0;
            const double x_diff = update - beta_k;
            WSYNC();
            if (lane == 0) beta[k] = update;
            for (int r = lane; r < pp; r += 64) mu[r] = fma(x_diff, (double)M[r + k * q], mu[r]);
            WSYNC();
        }
    }
    for (int i = lane; i < pp; i += 64) coeffs[i] = (T)beta[i];
}

template <typename T>
int launch_cd(pds_ctx* ctx, const T* d_moments, int p, int add_bias, double l1, double l2, double tol, int max_iter,
              int positive, T* d_coeffs, int* d_info, int64_t n_sys, uint8_t* d_flags, const int64_t* d_rows_per_sys) {
    if (n_sys <= 0) return PDS_OK;
    const size_t lds = (size_t)3 * (p + 2) * sizeof(double);
    if (lds > 160 * 1024) return fail(PDS_ERR_INVALID, "coordinate descent: more than 6800 features are not supported");
    if (lds > 48 * 1024)
        PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cd_kernel<T>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    KernelTimer timer(ctx, kKindIter);
    hipLaunchKernelGGL((cd_kernel<T>), dim3((unsigned)n_sys), dim3(64), lds, ctx->stream, d_moments, p, add_bias ? 1 : 0,
                       l1, l2, tol, max_iter, positive, d_coeffs, d_info, d_flags, d_rows_per_sys);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

template <typename T>
int launch_nnls(pds_ctx* ctx, const T* d_moments, int p, int add_bias, double tol, int max_iter, T* d_coeffs,
                int64_t n_sys, uint8_t* d_flags, const int64_t* d_rows_per_sys) {
    if (n_sys <= 0) return PDS_OK;
    const size_t lds = (size_t)2 * (p + 2) * sizeof(double);
    KernelTimer timer(ctx, kKindIter);
    hipLaunchKernelGGL((nnls_kernel<T>), dim3((unsigned)n_sys), dim3(64), lds, ctx->stream, d_moments, p,
                       add_bias ? 1 : 0, tol, max_iter, d_coeffs, d_flags, d_rows_per_sys);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

template int launch_solve<double>(pds_ctx*, const double*, int64_t, const SolveParams&, double*, uint8_t*, double*,
                                  const int64_t*);
template int launch_solve<float>(pds_ctx*, const float*, int64_t, const SolveParams&, float*, uint8_t*, float*,
                                 const int64_t*);
template int launch_cd<double>(pds_ctx*, const double*, int, int, double, double, double, int, int, double*, int*, int64_t,
                               uint8_t*, const int64_t*);
template int launch_cd<float>(pds_ctx*, const float*, int, int, double, double, double, int, int, float*, int*, int64_t,
                              uint8_t*, const int64_t*);
template int launch_nnls<double>(pds_ctx*, const double*, int, int, double, int, double*, int64_t, uint8_t*, const int64_t*);
template int launch_nnls<float>(pds_ctx*, const float*, int, int, double, int, float*, int64_t, uint8_t*, const int64_t*);

}  // namespace pds
