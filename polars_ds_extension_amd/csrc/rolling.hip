// rolling.hip -- pl_rolling_lr / pl_recursive_lr (/root/reference/src/num_ext/linear_regression.rs
// :1121-1283) without the reference's strictly sequential Sherman-Morrison-Woodbury chain
// (lr_online_solvers.rs:148-332: 2 rank-1 updates per row, one row after the other, 1e8 times).
//
// The window normal equations are rebuilt instead of the inverse being dragged along:
//     G(r) = sum_{i in window(r), row i finite} z_i z_i' ,  z = [x_0..x_{p-1}, (1)],   c(r) likewise,
//     (G(r) + lambda I) beta(r) = c(r)          (lambda on every diagonal)
// which is what the reference's recursion equals in exact arithmetic, without its drift.
//
// gfx950 mapping.  One wavefront owns a tile of kTileRows consecutive rows and walks it 64 rows at a time:
//   A  lane = row: coalesced 8-byte/lane loads of row r and row r-w, the NV = p'(p'+1)/2 + p' + 1 moment
//      increments d_v = z_a z_b (new) - z_a z_b (old) go to LDS as D[v][lane]          (stride 65: no conflicts)
//   B  lane = moment v: a 64-long running sum W_v(r) = W_v(r-1) + D[v][r] in place (carry kept in a VGPR)
//   C  lane = row again: reads its own G(r), c(r) back, Cholesky-solves the p' x p' system in registers
//      (fully unrolled), forms pred = x_r . beta, and stores the List<f64> values / pred / validity.
// A tile is anchored exactly: the carry is rebuilt from the w rows in front of the tile (rolling) or
// from an exclusive prefix over per-tile totals (expanding), so round-off never accumulates over more
// than kTileRows additions.  Non-finite rows are left out of the sums (OnlineLR::update :85-89) and, with
// min_size > 0, counted for the skipping variant's validity rule (faer_rolling_skipping_lr :218-301).
// HBM traffic: every input element read twice (as "new" and as "old", the second read served by
// L2/Infinity Cache for w = 256), every output element written once.
#include <type_traits>

#include "common.hpp"

namespace pds {

#define RSYNC() PDS_WAVE_LDS_SYNC()

// -DPDS_PROFILE_ROLLING: shader-clock sums per phase over all waves (development; tools/rolling_profile.py)
#ifdef PDS_PROFILE_ROLLING
__device__ unsigned long long g_roll_cycles[8];
#define RT0() unsigned long long _t0 = __builtin_amdgcn_s_memtime()
#define RTA() _t0 = __builtin_amdgcn_s_memtime()
#define RT1(k) rprof[k] += __builtin_amdgcn_s_memtime() - _t0
#else
#define RT0() do {} while (0)
#define RTA() do {} while (0)
#define RT1(k) do {} while (0)
#endif

#ifndef PDS_ROLL_WPE
#define PDS_ROLL_WPE 2
#endif
constexpr int kTileRows = 4096;   // rows per wave tile (64 steps of 64 rows)
constexpr int kRollWaves = 2;     // waves per block
constexpr int kLdsStride = 65;    // doubles per moment row in LDS

template <int PP>
struct RollDims {
    static constexpr int NG = PP * (PP + 1) / 2;
    static constexpr int NV = NG + PP + 1;  // Gram upper triangle, X'y, finite-row count
    // The moments go through LDS in NSET passes of NH moments each (moment v = pass v / NH, lane v % NH): one wave's
    // LDS stays under 20 KB, so 8 waves per CU fit (at p' = 8 all 45 moments at once were 23.4 KB -> 6 waves per CU,
    // and the step is latency bound: measured 1.5x between 6 and 8 waves per CU).
    // NH <= 32: a pass's moments are scanned by TWO lanes each (rows 0-31 and 32-63 of the step), see phase B.
    static constexpr int NSET = (NV + 31) / 32;
    static constexpr int NH = (NV + NSET - 1) / NSET;
};

struct RollArgs {
    int p, pp, bias;        // features, p + bias, bias flag
    int64_t n, window;      // window = n0 (start_with) for expanding
    int64_t min_size;
    double lambda;
    int mode;               // 0 rolling, 1 expanding pass 1 (tile totals), 2 expanding main
    int64_t tile_rows;
};

}  // namespace pds
#include "rolling_seg_dev.hpp"  // (needs RollArgs)
namespace pds {

// fetch_row: raw z (PP entries, padding = 0, bias entry = 1) and y of row r (zeros when r is out of range) -- only
// issues the loads, so the next step's rows can be in flight while the current step is scanned and solved;
// finish_row: the finiteness rule (a non-finite row is left out, OnlineLR::update lr_online_solvers.rs:85-89).
template <typename T, int PP>
__device__ __forceinline__ bool fetch_row(const T* const* __restrict__ cols, const RollArgs& ra, int64_t r,
                                          double (&z)[PP], double& yv) {
    const bool in = r >= 0 && r < ra.n;
#pragma unroll
    for (int a = 0; a < PP; ++a) {
        double v = 0.0;
        if (a < ra.p) v = in ? (double)as_global(cols[a])[r] : 0.0;
        else if (a == ra.p && ra.bias) v = 1.0;
        z[a] = v;
    }
    yv = in ? (double)as_global(cols[ra.p])[r] : 0.0;
    return in;
}
template <int PP>
__device__ __forceinline__ bool finish_row(bool in, const double (&z)[PP], double yv) {
    bool fin = isfinite(yv);
#pragma unroll
    for (int a = 0; a < PP; ++a) fin = fin && isfinite(z[a]);
    return in && fin;
}

// MODE (0 rolling, 1 expanding totals, 2 expanding main) and FULLP (1: p == PP without a bias column, 2: p == PP - 1 plus the bias column; 0: run time) are compile-time: the step
// is VALU bound, but the wave-uniform branches on them (a dozen per row fetch, more in the phases) each cost a scalar
// compare + branch bubble inside it.
template <typename T, int PP, int MODE, int FULLP>
__device__ __forceinline__ void rolling_body(const T* const* __restrict__ cols, const RollArgs& ra_in,
                                             double* __restrict__ tile_tot /*[tiles][NV]*/, T* __restrict__ coeffs,
                                             T* __restrict__ pred, uint8_t* __restrict__ valid) {
    RollArgs ra = ra_in;
    ra.mode = MODE;
    if constexpr (FULLP == 1) {
        ra.p = PP;
        ra.pp = PP;
        ra.bias = 0;
    } else if constexpr (FULLP == 2) {
        ra.p = PP - 1;
        ra.pp = PP;
        ra.bias = 1;
    }
    constexpr int NG = RollDims<PP>::NG, NV = RollDims<PP>::NV, NSET = RollDims<PP>::NSET, NH = RollDims<PP>::NH;
    static_assert(NH <= 32, "two lanes per moment of a pass");
    extern __shared__ __attribute__((aligned(16))) double seg_lds[];
    double* sm = seg_lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* D = sm + (size_t)wave * NH * kLdsStride;
    const int64_t T_ = ra.tile_rows;
    const int64_t ntiles = (ra.n + T_ - 1) / T_;
    const int64_t wid = (int64_t)blockIdx.x * kRollWaves + wave, nw = (int64_t)gridDim.x * kRollWaves;
    const int64_t w = ra.window;

#ifdef PDS_PROFILE_ROLLING
    unsigned long long rprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    for (int64_t t = wid; t < ntiles; t += nw) {
        const int64_t t0 = t * T_, t1 = (t0 + T_ < ra.n) ? t0 + T_ : ra.n;
        // running moments: lane l < NH owns moment k * NH + l of every pass k
        double W[NSET];
#pragma unroll
        for (int k = 0; k < NSET; ++k) W[k] = 0.0;
        int64_t r_begin = t0;
        if (ra.mode == 0) {
            r_begin = t0 - ((w + 63) / 64) * 64;  // warm-up steps rebuild the window in front of the tile
        } else if (ra.mode == 2) {
#pragma unroll
            for (int k = 0; k < NSET; ++k)
                if (lane < NH && k * NH + lane < NV) W[k] = tile_tot[t * NV + k * NH + lane];  // exclusive prefix over the previous tiles
        }
        // rows of the first step
        double nn[PP], no[PP], nyn, nyo = 0.0;
        bool nin_n = fetch_row<T, PP>(cols, ra, r_begin + lane, nn, nyn), nin_o = false;
        if (ra.mode == 0 && r_begin >= t0) nin_o = fetch_row<T, PP>(cols, ra, r_begin + lane - w, no, nyo);
        for (int64_t base = r_begin; base < t1; base += 64) {
            const bool warm = base < t0;
            const int64_t r = base + lane;
            // ---------------- phase A
            RT0();
            double zn[PP], zo[PP], yn = nyn, yo = nyo;
#pragma unroll
            for (int a = 0; a < PP; ++a) {
                zn[a] = nn[a];
                zo[a] = no[a];
            }
            bool okn = finish_row<PP>(nin_n, zn, yn);
            if (warm) okn = okn && (r >= t0 - w);          // only the w rows in front of the tile
            else okn = okn && (r < t1);
            bool oko = false;
            if (ra.mode == 0 && !warm) oko = finish_row<PP>(nin_o, zo, yo) && (r < t1);
            // next step's rows go in flight now; they are consumed after this step's scan and solve
            {
                const int64_t nb = base + 64;
                if (nb < t1) {
                    nin_n = fetch_row<T, PP>(cols, ra, nb + lane, nn, nyn);
                    nin_o = false;
                    if (ra.mode == 0 && nb >= t0) nin_o = fetch_row<T, PP>(cols, ra, nb + lane - w, no, nyo);
                }
            }
            if (!okn) {
#pragma unroll
                for (int a = 0; a < PP; ++a) zn[a] = 0.0;
                yn = 0.0;
            }
            if (!oko) {
#pragma unroll
                for (int a = 0; a < PP; ++a) zo[a] = 0.0;
                yo = 0.0;
            }
            RT1(0);  // row hand-over: wait for the prefetched rows, finiteness, issue of the next rows
            double g[NG], c[PP], cnt = 0.0;
            const bool want_c = !(warm || ra.mode == 1);
#pragma unroll
            for (int ks = 0; ks < NSET; ++ks) {
                // ---- A: the increments of this pass's moments, D[moment % NH][row]
                {
                    RT0();
                    int v = 0;
#pragma unroll
                    for (int a = 0; a < PP; ++a)
#pragma unroll
                        for (int b = a; b < PP; ++b) {
                            if (v / NH == ks) D[(v % NH) * kLdsStride + lane] = fma(zn[a], zn[b], -(zo[a] * zo[b]));
                            ++v;
                        }
#pragma unroll
                    for (int a = 0; a < PP; ++a)
                        if ((NG + a) / NH == ks) D[((NG + a) % NH) * kLdsStride + lane] = fma(zn[a], yn, -(zo[a] * yo));
                    if ((NG + PP) / NH == ks) D[((NG + PP) % NH) * kLdsStride + lane] = (okn ? 1.0 : 0.0) - (oko ? 1.0 : 0.0);
#ifdef PDS_PROFILE_ROLLING
                    RSYNC();
#endif
                    RT1(1);
                }
                RSYNC();
                // ---- B: lane v scans its moment over the 64 rows of this step
                RT0();
                {
                    // Two lanes per moment: lane m scans rows 0-31 from the carry, lane 32 + m rows 32-63 from zero, and
                    // the total of the first half (carry included) goes to the row's spare LDS slot [64], which phase C
                    // adds for the rows of the upper half.  Half the LDS instructions and half the dependent chain of
                    // one lane per moment (the scan was 44 % of the kernel's time).  Per batch: 16 independent LDS
                    // reads, 16 dependent adds, 16 writes (a read-add-write loop would put an LDS round trip into
                    // every link of the chain).
                    const int hh = lane >> 5, m = lane & 31;
                    double run = 0.0;
                    if (m < NH && ks * NH + m < NV) {
                        double* row = D + m * kLdsStride + 32 * hh;
                        run = (hh == 0) ? W[ks] : 0.0;
#pragma unroll
                        for (int i0 = 0; i0 < 32; i0 += 16) {
                            double v[16];
#pragma unroll
                            for (int i = 0; i < 16; ++i) v[i] = row[i0 + i];
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                run += v[i];
                                v[i] = run;
                            }
#pragma unroll
                            for (int i = 0; i < 16; ++i) row[i0 + i] = v[i];
                        }
                        if (hh == 0) row[64] = run;
                    }
                    // carry of the next step = first half's total (carry included) + second half's local total
                    const unsigned lo = (unsigned)__double2loint(run), hi = (unsigned)__double2hiint(run);
                    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
                    const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
                    W[ks] = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
                }
                RSYNC();
                RT1(2);
                // ---- C (first half): lane = row again takes its window sums of this pass into registers
                if (want_c) {
                    const double up = (lane >= 32) ? 1.0 : 0.0;  // rows of the upper half add the first half's total
                    int v = 0;
#pragma unroll
                    for (int a = 0; a < PP; ++a)
#pragma unroll
                        for (int b = a; b < PP; ++b) {
                            if (v / NH == ks) {
                                double x = fma(up, D[(v % NH) * kLdsStride + 64], D[(v % NH) * kLdsStride + lane]);
                                if (a == b) {
                                    if (a < ra.pp) x += ra.lambda;
                                    else x = 1.0;  // padding dimension: identity, beta_pad = 0
                                }
                                g[v] = x;
                            }
                            ++v;
                        }
#pragma unroll
                    for (int a = 0; a < PP; ++a)
                        if ((NG + a) / NH == ks) c[a] = fma(up, D[((NG + a) % NH) * kLdsStride + 64], D[((NG + a) % NH) * kLdsStride + lane]);
                    if ((NG + PP) / NH == ks) cnt = fma(up, D[((NG + PP) % NH) * kLdsStride + 64], D[((NG + PP) % NH) * kLdsStride + lane]);
                }
                if (ks + 1 < NSET) RSYNC();  // the next pass overwrites D
            }
            if (!want_c) continue;
            // ---------------- phase C: lane = row, solve (G + lambda I) beta = c
            RTA();
            if (r < t1) {
                // Cholesky G = L L' in place (packed upper storage read as lower by symmetry):
                // idx(a,b), a <= b  ->  a*PP - a(a-1)/2 + (b-a)
                bool okc = true;
                double ri[PP];  // 1 / l_kk: hardware rsqrt estimate + two Newton steps; the substitutions multiply by it
                                // (sqrt + 24 divisions per row were ~3/4 of this kernel's VALU instructions)
#define GI(a, b) g[(a) * PP - ((a) * ((a)-1)) / 2 + ((b) - (a))]
#pragma unroll
                for (int k = 0; k < PP; ++k) {
                    const double d = GI(k, k);
                    okc = okc && (d > 0.0);
                    double inv = __builtin_amdgcn_rsq(d);
                    inv = inv * fma(-0.5 * d * inv, inv, 1.5);
                    inv = inv * fma(-0.5 * d * inv, inv, 1.5);
                    ri[k] = inv;
#pragma unroll
                    for (int b = k + 1; b < PP; ++b) GI(k, b) *= inv;  // l_bk stored at (k,b)
#pragma unroll
                    for (int a = k + 1; a < PP; ++a)
#pragma unroll
                        for (int b = a; b < PP; ++b) GI(a, b) = fma(-GI(k, a), GI(k, b), GI(a, b));
                }
                // forward L u = c, backward L' beta = u
#pragma unroll
                for (int a = 0; a < PP; ++a) {
                    double s = c[a];
#pragma unroll
                    for (int k = 0; k < a; ++k) s = fma(-GI(k, a), c[k], s);
                    c[a] = s * ri[a];
                }
#pragma unroll
                for (int a = PP - 1; a >= 0; --a) {
                    double s = c[a];
#pragma unroll
                    for (int k = a + 1; k < PP; ++k) s = fma(-GI(a, k), c[k], s);
                    c[a] = s * ri[a];
                }
#undef GI
                const double nanv = __builtin_nan("");
                const bool row_ok = (ra.mode == 0) ? (r >= w - 1) : (r >= w - 1);
                bool v_ok = row_ok;
                if (ra.min_size > 0) v_ok = v_ok && (cnt >= (double)ra.min_size);
                double pr = 0.0;
#pragma unroll
                for (int a = 0; a < PP; ++a) pr = fma(zn[a], c[a], pr);
                if (!okn) pr = nanv;  // the row itself is non-finite: x_r . beta is NaN in the reference too
                T* out = coeffs + r * (int64_t)ra.pp;
#pragma unroll
                for (int a = 0; a < PP; ++a)
                    if (a < ra.pp) out[a] = (v_ok && okc) ? (T)c[a] : (T)nanv;
                pred[r] = (v_ok && okc) ? (T)pr : (T)nanv;
                valid[r] = v_ok ? 1 : 0;
            }
            RSYNC();
            RT1(3);
        }
        if (ra.mode == 1) {
#pragma unroll
            for (int k = 0; k < NSET; ++k)
                if (lane < NH && k * NH + lane < NV) tile_tot[t * NV + k * NH + lane] = W[k];
        }
    }
#ifdef PDS_PROFILE_ROLLING
    rprof[7] = __builtin_amdgcn_s_memtime() - t_begin;
    if (lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_roll_cycles[k], rprof[k]);
#endif
}

// p' <= 8 is compiled for two waves per SIMD (248 VGPRs, no spills): with the moments passing through LDS in halves
// the CU then holds 8 waves.  p' >= 10 would spill at that budget (measured 2x slower) and keeps one wave per SIMD.
template <typename T, int PP, int MODE, int FULLP>
__global__ __launch_bounds__(kRollWaves * 64) __attribute__((amdgpu_waves_per_eu(PDS_ROLL_WPE))) void rolling_kernel(
    const T* const* __restrict__ cols, RollArgs ra, double* __restrict__ tile_tot, T* __restrict__ coeffs, T* __restrict__ pred,
    uint8_t* __restrict__ valid) {
    rolling_body<T, PP, MODE, FULLP>(cols, ra, tile_tot, coeffs, pred, valid);
}
template <typename T, int PP, int MODE, int FULLP>
__global__ __launch_bounds__(kRollWaves * 64) void rolling_kernel_1w(const T* const* __restrict__ cols, RollArgs ra,
                                                                     double* __restrict__ tile_tot, T* __restrict__ coeffs,
                                                                     T* __restrict__ pred, uint8_t* __restrict__ valid) {
    rolling_body<T, PP, MODE, FULLP>(cols, ra, tile_tot, coeffs, pred, valid);
}
template <typename T, int PP, int MODE, int FULLP>
static auto roll_kernel_ptr() {
    if constexpr (PP >= 10) return &rolling_kernel_1w<T, PP, MODE, FULLP>;
    else return &rolling_kernel<T, PP, MODE, FULLP>;
}

// exclusive prefix over the per-tile totals (expanding window) in three small launches: chunk sums (one wave per chunk of
// kPrefixChunk tiles, lane = moment), the running bases of the chunks (one wave, chunk order), the prefixes inside every chunk.
// A fixed summation order -- chunk by chunk, tile by tile -- so results do not depend on scheduling.  (One 16-wave block
// walking all 24 414 tiles of a 1e8-row frame twice took 0.62 ms.)
// (`seed`: moments of rows that precede this frame -- the row-sharded multi-GPU expanding fit -- or nullptr)
constexpr int kPrefixChunk = 32;
__global__ __launch_bounds__(64) void tile_chunk_sum_kernel(const double* __restrict__ tot, int64_t ntiles, int nv,
                                                            double* __restrict__ chunk_sum) {
    const int64_t c = blockIdx.x;
    const int64_t t0 = c * kPrefixChunk, t1 = (t0 + kPrefixChunk < ntiles) ? t0 + kPrefixChunk : ntiles;
    for (int v = threadIdx.x; v < nv; v += 64) {
        double sum = 0.0;
        for (int64_t t = t0; t < t1; ++t) sum += tot[t * nv + v];
        chunk_sum[c * nv + v] = sum;
    }
}
__global__ __launch_bounds__(128) void chunk_prefix_kernel(double* __restrict__ chunk_sum, int64_t nchunks, int nv,
                                                           const double* __restrict__ seed) {
    const int v = threadIdx.x;
    if (v >= nv) return;
    double run = seed ? seed[v] : 0.0;
    int64_t c = 0;
    for (; c + 8 <= nchunks; c += 8) {
        double x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = chunk_sum[(c + k) * nv + v];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            chunk_sum[(c + k) * nv + v] = run;
            run += x[k];
        }
    }
    for (; c < nchunks; ++c) {
        const double x = chunk_sum[c * nv + v];
        chunk_sum[c * nv + v] = run;
        run += x;
    }
}
__global__ __launch_bounds__(64) void tile_prefix_write_kernel(double* __restrict__ tot, int64_t ntiles, int nv,
                                                               const double* __restrict__ chunk_base) {
    const int64_t c = blockIdx.x;
    const int64_t t0 = c * kPrefixChunk, t1 = (t0 + kPrefixChunk < ntiles) ? t0 + kPrefixChunk : ntiles;
    for (int v = threadIdx.x; v < nv; v += 64) {
        double run = chunk_base[c * nv + v];
        for (int64_t t = t0; t < t1; ++t) {
            const double x = tot[t * nv + v];
            tot[t * nv + v] = run;
            run += x;
        }
    }
}
static int launch_tile_prefix(pds_ctx* ctx, double* tot, int64_t ntiles, int nv, const double* d_seed) {
    if (nv > 128) return fail(PDS_ERR_INVALID, "internal: tile prefix handles up to 128 moments");
    const int64_t nchunks = (ntiles + kPrefixChunk - 1) / kPrefixChunk;
    double* chunk = reinterpret_cast<double*>(ws_take(ctx, (size_t)nchunks * nv * sizeof(double)));
    if (!chunk) return fail(PDS_ERR_HIP, "workspace allocation failed");
    hipLaunchKernelGGL(tile_chunk_sum_kernel, dim3((unsigned)nchunks), dim3(64), 0, ctx->stream, (const double*)tot, ntiles, nv, chunk);
    hipLaunchKernelGGL(chunk_prefix_kernel, dim3(1), dim3(128), 0, ctx->stream, chunk, nchunks, nv, d_seed);
    hipLaunchKernelGGL(tile_prefix_write_kernel, dim3((unsigned)nchunks), dim3(64), 0, ctx->stream, tot, ntiles, nv, (const double*)chunk);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

template <typename T, int PP, int FULLP>
static int launch_pp_f(pds_ctx* ctx, const DeviceCols<T>& dc, RollArgs ra, bool expanding, const double* seed_moments,
                       T* d_coeffs, T* d_pred, uint8_t* d_valid) {
    constexpr int NV = RollDims<PP>::NV;
    const size_t lds = (size_t)kRollWaves * RollDims<PP>::NH * kLdsStride * sizeof(double);
    ra.tile_rows = kTileRows;
    const int64_t ntiles = (ra.n + ra.tile_rows - 1) / ra.tile_rows;
    int64_t nb = (ntiles + kRollWaves - 1) / kRollWaves;
    int per_cu = 2 * PDS_ROLL_WPE;  // blocks of kRollWaves waves: PDS_ROLL_WPE waves per SIMD
    if (const char* e = dev_env("PDS_ROLL_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(e));
    nb = std::min<int64_t>(std::max<int64_t>(nb, 1), (int64_t)ctx->num_cus * per_cu);
    auto kern0 = roll_kernel_ptr<T, PP, 0, FULLP>();
    auto kern1 = roll_kernel_ptr<T, PP, 1, FULLP>();
    auto kern2 = roll_kernel_ptr<T, PP, 2, FULLP>();
    if (lds > 64 * 1024)
        for (const void* k : {reinterpret_cast<const void*>(kern0), reinterpret_cast<const void*>(kern1), reinterpret_cast<const void*>(kern2)})
            PDS_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // up to 8 coefficients: lane = 4 consecutive rows, running moments in registers (rolling_seg_dev.hpp); PDS_ROLLING_V1=1
    // keeps the lane = row kernel (development A/B).  The totals pass of the expanding fit stays with the first kernel.
    static const bool v1 = [] { const char* e = dev_env("PDS_ROLLING_V1"); return e && e[0] == '1'; }();
    bool seg = false;
    if constexpr (PP <= 8) seg = !v1;
    [[maybe_unused]] auto launch_seg = [&](auto mode_c, const double* tot) {
        if constexpr (PP <= 8) {
            constexpr int M = decltype(mode_c)::value;
            constexpr int64_t tile = M == 0 ? kSegTileRoll : kSegTile;
            const int64_t seg_tiles = (ra.n + tile - 1) / tile;
            const int64_t sb = std::min<int64_t>(std::max<int64_t>(seg_tiles, 1), (int64_t)ctx->num_cus * 4);
            // window == one stage: the rows leaving the window are the previous stage's rows, kept in registers (no second read
            // stream); PDS_ROLL_OLD_STREAM=1 keeps the two-stream form for that window too (A/B)
            static const bool old_stream = [] { const char* e = dev_env("PDS_ROLL_OLD_STREAM"); return e && e[0] == '1'; }();
            if (M == 0 && ra.window == kSegStage && !old_stream) {
                if constexpr (M == 0) {
                    using SD = SegDims<T, PP, 1>;
                    hipLaunchKernelGGL((rolling_seg_kernel<T, PP, 0, FULLP, 1>), dim3((unsigned)sb), dim3(64), (size_t)SD::LDS_BYTES,
                                       ctx->stream, dc.d_ptrs, ra, tot, d_coeffs, d_pred, d_valid);
                }
            } else {
                using SD = SegDims<T, PP, 0>;
                hipLaunchKernelGGL((rolling_seg_kernel<T, PP, M, FULLP, 0>), dim3((unsigned)sb), dim3(64), (size_t)SD::LDS_BYTES,
                                   ctx->stream, dc.d_ptrs, ra, tot, d_coeffs, d_pred, d_valid);
            }
        }
    };
    KernelTimer timer(ctx, kKindRolling);
    if (!expanding) {
        ra.mode = 0;
        if (seg) launch_seg(std::integral_constant<int, 0>{}, (const double*)nullptr);
        else
            hipLaunchKernelGGL(kern0, dim3((unsigned)nb), dim3(kRollWaves * 64), lds, ctx->stream,
                               dc.d_ptrs, ra, (double*)nullptr, d_coeffs, d_pred, d_valid);
    } else {
        double* tot = reinterpret_cast<double*>(ws_take(ctx, (size_t)ntiles * NV * sizeof(double)));
        double* d_seed = nullptr;
        if (seed_moments) {
            // augmented (p+2)^2 moment matrix (pds_moments layout, Z = [x | 1 | y]) -> this kernel's moment vector:
            // upper triangle of z z' with z = [x, (1)], then z y, then the row count
            const int p = ra.p, q = ra.p + 2;
            double h[RollDims<PP>::NV];
            auto zi = [&](int a) { return a < p ? a : p; };
            int v = 0;
            for (int a = 0; a < PP; ++a)
                for (int b = a; b < PP; ++b) h[v++] = (a < ra.pp && b < ra.pp) ? seed_moments[zi(a) + (size_t)zi(b) * q] : 0.0;
            for (int a = 0; a < PP; ++a) h[v++] = (a < ra.pp) ? seed_moments[zi(a) + (size_t)(p + 1) * q] : 0.0;
            h[v++] = seed_moments[p + (size_t)p * q];
            d_seed = reinterpret_cast<double*>(ws_take(ctx, NV * sizeof(double)));
            PDS_HIP_CHECK(hipMemcpyAsync(d_seed, h, NV * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // h is on this stack frame
        }
        ra.mode = 1;
        bool totals_done = false;
        if constexpr (PP <= 8) {
            if (seg) {  // streaming totals (rolling_seg_dev.hpp): HBM bound instead of a full pass of the rolling kernel
                using SD = SegDims<T, PP>;
                const int64_t tb = std::min<int64_t>(std::max<int64_t>(ntiles, 1), (int64_t)ctx->num_cus * 12);
                hipLaunchKernelGGL((rolling_totals_kernel<T, PP, FULLP>), dim3((unsigned)tb), dim3(64),
                                   (size_t)SD::NV * kSegStride * sizeof(double), ctx->stream, dc.d_ptrs, ra, tot);
                totals_done = true;
            }
        }
        if (!totals_done)
            hipLaunchKernelGGL(kern1, dim3((unsigned)nb), dim3(kRollWaves * 64), lds, ctx->stream,
                               dc.d_ptrs, ra, tot, d_coeffs, d_pred, d_valid);
        if (int rc = launch_tile_prefix(ctx, tot, ntiles, NV, d_seed)) return rc;
        ra.mode = 2;
        if (seg) launch_seg(std::integral_constant<int, 2>{}, (const double*)tot);
        else
            hipLaunchKernelGGL(kern2, dim3((unsigned)nb), dim3(kRollWaves * 64), lds, ctx->stream,
                               dc.d_ptrs, ra, tot, d_coeffs, d_pred, d_valid);
    }
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

template <typename T, int PP>
static int launch_pp(pds_ctx* ctx, const DeviceCols<T>& dc, RollArgs ra, bool expanding, const double* seed_moments,
                     T* d_coeffs, T* d_pred, uint8_t* d_valid) {
    if (ra.p == PP && !ra.bias) return launch_pp_f<T, PP, 1>(ctx, dc, ra, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    if (ra.p == PP - 1 && ra.bias) return launch_pp_f<T, PP, 2>(ctx, dc, ra, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    return launch_pp_f<T, PP, 0>(ctx, dc, ra, expanding, seed_moments, d_coeffs, d_pred, d_valid);
}

template <typename T>
int launch_rolling(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, int add_bias, int64_t window,
                   int64_t min_size, double lambda, bool expanding, const double* seed_moments, T* d_coeffs, T* d_pred,
                   uint8_t* d_valid) {
    RollArgs ra;
    ra.p = n_feat;
    ra.bias = add_bias ? 1 : 0;
    ra.pp = n_feat + ra.bias;
    ra.n = n_rows;
    ra.window = window;
    ra.min_size = min_size;
    ra.lambda = lambda > 0.0 ? lambda : 0.0;
    ra.mode = 0;
    ra.tile_rows = kTileRows;
    const int pp = ra.pp;
    if (pp <= 2) return launch_pp<T, 2>(ctx, dc, ra, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    if (pp <= 4) return launch_pp<T, 4>(ctx, dc, ra, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    if (pp <= 6) return launch_pp<T, 6>(ctx, dc, ra, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    if (pp <= 8) return launch_pp<T, 8>(ctx, dc, ra, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    if (pp <= 10) return launch_pp<T, 10>(ctx, dc, ra, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    if (pp <= 12) return launch_pp<T, 12>(ctx, dc, ra, expanding, seed_moments, d_coeffs, d_pred, d_valid);
    // 13 .. 64 coefficients: per-row moment records + the batched solver (rolling_wide.hip)
    return launch_rolling_wide<T>(ctx, dc, n_feat, n_rows, add_bias, window, min_size, lambda, expanding, seed_moments, d_coeffs,
                                  d_pred, d_valid);
}

#ifdef PDS_PROFILE_ROLLING
extern "C" int pds_debug_rolling_cycles(unsigned long long* out, int reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_roll_cycles), sizeof(z)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_roll_cycles), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

template int launch_rolling<double>(pds_ctx*, const DeviceCols<double>&, int, int64_t, int, int64_t, int64_t, double,
                                    bool, const double*, double*, double*, uint8_t*);
template int launch_rolling<float>(pds_ctx*, const DeviceCols<float>&, int, int64_t, int, int64_t, int64_t, double, bool,
                                   const double*, float*, float*, uint8_t*);

}  // namespace pds
