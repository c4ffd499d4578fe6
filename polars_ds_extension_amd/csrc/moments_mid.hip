// moments_mid.hip -- the Gram / moment build of ONE regression with 17 .. 64 f64 features: a STREAMING kernel in the style of the
// p <= 16 kernel (moments.hip), several 16-feature tile columns wide.
//
// get_xtx_with_lambda + build_xty (lr_solvers.rs:183-211, 262-278) for the usual width of a hand-built regression.  The tiled
// SYRK of moments_wide.hip serves these widths through a compact form (MODE 3 / 4 / 5: one 128-column block, tiles dealt to four
// waves, a shared LDS panel, two barriers per 32-row stage, ONE stage of loads in flight): 1.5 - 2.0 TB/s -- a 17th feature cost
// four times the 16-feature kernel per row.  Here nothing is shared and nothing waits for a neighbour:
//   * one wave per SIMD owns a contiguous row range and a private LDS image of two half-tiles of HR rows x (16 NBLK + 1) columns;
//     the next half-tile is fetched global -> LDS asynchronously (global_load_lds, 16 bytes per lane straight down a column) while
//     the matrix core consumes the current one;
//   * per 4-row step the wave reads NBLK operand registers (lane = (feature i of the block, row slot k): column stride HR * 8 + 16
//     bytes keeps the half-wave's reads on distinct banks) and issues NBLK (NBLK + 1) / 2 v_mfma_f64_16x16x4_f64 -- the upper
//     block triangle of X'X, A and B operands being the same registers -- with independent accumulators;
//   * X'y, the column sums, y'y and sum y ride on the VALU from the same operand registers (the target is one more LDS column);
//   * per-wave partial records, summed in wave order by a finalize kernel: no atomics, bit-reproducible.
// Bound: HBM up to 32 features (3 matrix instructions per 4 rows = 0.52 ms per 2e7 rows against 1.05 ms of HBM time for
// 2e7 x 33 doubles), the f64 matrix pipe beyond (10 instructions per 4 rows at 64 features: 1.75 ms of pipe per 2e7 rows at the
// measured 86 clk per instruction and SIMD, tools/mfma_peak.hip, against 1.30 ms of HBM time).
#include "common.hpp"
#include "moments_dev.hpp"
#include "moments_mid_dev.hpp"

namespace pds {

namespace {

// WEIGHTED: cols[p + 1] = w; the record is then Z' diag(w) Z (faer_weighted_lr's X'WX | X'Wy, lr_solvers.rs:386-409; entry (1, 1) = sum w)
// FUSE (the second pass of pl_lin_reg_report's robust errors, linear_regression.rs:880-909, in ONE stream over the frame): the weights
// are made in the kernel from the half-tile it has just loaded -- s_r = e_r^2 (FUSE = 1: HC0 / HC1) or e_r^2 / (1 - h_r)^(hc - 1)
// (FUSE = 2: HC2 / HC3, h_r = ||L' z_r||^2 on the matrix cores as in leverage_mid.hip) -- written to the weight column's LDS image and
// consumed by the weighted Gram steps: the record is the meat X' diag(s) X, sum e^2 rides along, and neither the residuals nor the
// n-row weight vector ever exist in memory.
template <int NBLK, bool WEIGHTED, int FUSE, typename T = double>
__global__ __launch_bounds__(FUSE ? 256 : 64) void moments_mid_kernel(const T* const* __restrict__ cols, int p, int64_t n,
                                                                      double* __restrict__ partials, int bias, const double* __restrict__ beta,
                                                                      const double* __restrict__ lop, int hc) {
    constexpr int ES = (int)sizeof(T), EPL = 16 / ES;  // element bytes; elements per 16-byte lane piece
    static_assert(FUSE == 0 || ES == 8, "the fused report form is the f64 kernel's");
    using MD = MidDims<NBLK, ES>;
    using MS = MidShared<NBLK, FUSE>;
    constexpr int HR = MD::HR, GS = MD::GS, NPAIR = MD::NPAIR, NS = MD::NS;
    constexpr bool WT = WEIGHTED || FUSE != 0;      // the Gram steps read a weight image
    constexpr bool WLOAD = WEIGHTED && FUSE == 0;   // ... which comes from memory
    extern __shared__ __attribute__((aligned(16))) char mid_lds[];
    typedef __attribute__((address_space(3))) char* lds_c;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    typedef double mid_d2 __attribute__((ext_vector_type(2)));
#define PDS_MID_LDSD(addr) (*(__attribute__((address_space(3))) double*)(addr))
    const lds_c shared = (lds_c)mid_lds;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    lds_c sm = shared + MS::BYTES + wv * MD::LDS_BYTES;
    const int64_t wave = (int64_t)blockIdx.x * MS::WPB + wv, nwaves = (int64_t)gridDim.x * MS::WPB;
    if constexpr (FUSE != 0) {
        if constexpr (FUSE == 2)
            for (int i = threadIdx.x; i < MS::BETA_OFF / 8; i += 256) PDS_MID_LDSD(shared + i * 8) = lop[i];
        for (int i = threadIdx.x; i < 16 * NBLK + 2; i += 256)
            PDS_MID_LDSD(shared + MS::BETA_OFF + i * 8) = i < p ? beta[i] : ((i == 16 * NBLK && bias) ? beta[p] : 0.0);
    }
    // whole half-tiles, dealt as contiguous ranges; the ragged tail (n mod HR rows) belongs to the last wave
    const int64_t nh = n / HR;
    const int64_t h0 = nh * wave / nwaves, h1 = nh * (wave + 1) / nwaves;
    const int tail = (wave == nwaves - 1) ? (int)(n - nh * HR) : 0;
    // ---- the lane's 16 column pointers (column 16 g + i, g = its lane group), already advanced to its two rows of a half-tile;
    // `valid` bit i: that column exists (the others keep the zeros both images start with)
    const int g = lane / MD::GL, piece = lane % MD::GL;
    const T* cbase[16];
    unsigned valid = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = 16 * g + i;
        cbase[i] = cols[c < p ? c : p] + EPL * piece;
        if (c < p) valid |= 1u << i;
    }
    const T* ybase = cols[p] + EPL * lane;  // (lanes 0 .. GL - 1)
    const T* wbase = WLOAD ? cols[p + 1] + EPL * lane : ybase;
    for (int i = lane * 16; i < MD::LDS_BYTES; i += 64 * 16) *(__attribute__((address_space(3))) mid_d2*)(sm + i) = mid_d2{0.0, 0.0};
    static_assert(MD::LDS_BYTES % 16 == 0 && MS::BYTES % 16 == 0, "zeroed in 16-byte pieces");
    if constexpr (FUSE != 0) __syncthreads();  // (the only workgroup barrier: every wave reaches it)
    else PDS_WAVE_LDS_SYNC();
    // instruction i of a half-tile (i == 16: the target)
    auto issue_one = [&](int i, int buf, int64_t row0) __attribute__((always_inline)) {
        if (i < 16) {
            if ((valid >> i) & 1u)
                __builtin_amdgcn_global_load_lds((glb_ptr)(as_global(cbase[i]) + row0), (lds_ptr)(sm + buf * MD::HALF_BYTES + i * GS), 16, 0, 0);
        } else if (lane < MD::GL) {
            __builtin_amdgcn_global_load_lds((glb_ptr)(as_global(ybase) + row0), (lds_ptr)(sm + buf * MD::HALF_BYTES + MD::Y_OFF), 16, 0, 0);
            if constexpr (WLOAD)
                __builtin_amdgcn_global_load_lds((glb_ptr)(as_global(wbase) + row0), (lds_ptr)(sm + buf * MD::HALF_BYTES + MD::W_OFF), 16, 0, 0);
        }
    };
    auto issue = [&](int buf, int64_t row0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i <= 16; ++i) issue_one(i, buf, row0);
    };
    // guarded form for the ragged tail: zero rows beyond `rows`
    auto load_tail = [&](int buf, int64_t row0, int rows) __attribute__((always_inline)) {
        for (int c = 0; c <= p + (WLOAD ? 1 : 0); ++c) {
            const int off = c < p ? (c % 16) * GS + (c / 16) * HR * ES : (c == p ? MD::Y_OFF : MD::W_OFF);
            const gptr<T> col = as_global(cols[c]);
            for (int r = lane; r < HR; r += 64)
                *(__attribute__((address_space(3))) T*)(sm + buf * MD::HALF_BYTES + off + r * ES) = r < rows ? col[row0 + r] : T(0);
        }
    };
    d4 acc[NPAIR];
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
    double xy[NBLK], cs[NBLK], yy = 0.0, ys = 0.0, sw = 0.0;
#pragma unroll
    for (int b = 0; b < NBLK; ++b) xy[b] = cs[b] = 0.0;
    const int fi = lane & 15, fk = lane >> 4;
    // ---- fused report form: the weight image of half-tile `buf` (rows beyond `rows` get weight 0).  Per 16 rows: lane = (row fi,
    // feature slot fk); the residual's dot product rides on the operand registers of the first output block; the next half-tile's
    // loads go out between the row groups (they have the whole Gram phase to land)
    double sse = 0.0;
    auto produce = [&](int buf, int rows, bool next, int64_t next_row0) __attribute__((always_inline)) {
        const lds_c base = sm + buf * MD::HALF_BYTES;
        constexpr int RG = HR / 16, KS = 4 * NBLK, LPG = (17 + RG - 1) / RG;
        const double b0 = PDS_MID_LDSD(shared + MS::BETA_OFF + 16 * NBLK * 8);
        const double c0 = FUSE == 2 ? PDS_MID_LDSD(shared + MS::C0_OFF) : 0.0;
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            if (next) {
#pragma unroll
                for (int cc = 0; cc < LPG; ++cc)
                    if (rg * LPG + cc <= 16) issue_one(rg * LPG + cc, buf ^ 1, next_row0);
            }
            double predp = 0.0, hrow = 0.0;
            auto operand = [&](int ks) __attribute__((always_inline)) {
                return PDS_MID_LDSD(base + (4 * (ks & 3) + fk) * GS + (ks >> 2) * HR * 8 + (16 * rg + fi) * 8);
            };
            if constexpr (FUSE == 2) {
                double hs[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ablk = 0; ablk < NBLK; ++ablk) {
                    const double init = PDS_MID_LDSD(shared + MS::LB_OFF + (16 * ablk + fi) * 8);  // the intercept's row of L (zeros without)
                    d4 t = d4{init, init, init, init};
#pragma unroll
                    for (int ks = 4 * ablk; ks < KS; ++ks) {
                        const double a = operand(ks);
                        const double b = PDS_MID_LDSD(shared + mid_lev_block<NBLK>(ablk, ks) * 512 + lane * 8);
                        if (ablk == 0) predp = fma(a, PDS_MID_LDSD(shared + MS::BETA_OFF + (4 * ks + fk) * 8), predp);
                        t = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, t, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) hs[r] = fma(t[r], t[r], hs[r]);
                }
                // D has col = lane & 15 = output a, row = (lane >> 4) + 4 r: sum over a, then hand row fi its value
                double hv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = hs[r];
                    v += __shfl_xor(v, 1);
                    v += __shfl_xor(v, 2);
                    v += __shfl_xor(v, 4);
                    v += __shfl_xor(v, 8);
                    hv[r] = __shfl(v, 16 * (fi & 3));
                }
                const int rsel = fi >> 2;
                hrow = (rsel == 0 ? hv[0] : rsel == 1 ? hv[1] : rsel == 2 ? hv[2] : hv[3]) + c0;
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) predp = fma(operand(ks), PDS_MID_LDSD(shared + MS::BETA_OFF + (4 * ks + fk) * 8), predp);
            }
            predp += __shfl_xor(predp, 16);
            predp += __shfl_xor(predp, 32);
            const int lr = 16 * rg + fi;
            const double e = PDS_MID_LDSD(base + MD::Y_OFF + lr * 8) - (predp + b0);
            const double e2 = lr < rows ? e * e : 0.0;
            double sc = 1.0;
            if constexpr (FUSE == 2) {
                const double om = 1.0 - hrow;
                sc = hc == 2 ? 1.0 / om : 1.0 / (om * om);
            }
            if (fk == 0) {
                PDS_MID_LDSD(base + MD::W_OFF + lr * 8) = lr < rows ? e2 * sc : 0.0;
                sse += e2;
            }
        }
    };
    // `next`: the following half-tile's 17 loads are issued between the matrix instructions of the first steps
    constexpr int IPS = (17 + NS - 1) / NS < 2 ? 2 : (17 + NS - 1) / NS;  // load instructions per step
    auto consume = [&](int buf, bool next, int64_t next_row0) __attribute__((always_inline)) {
        const lds_c base = sm + buf * MD::HALF_BYTES;
        auto fetch = [&](int s, double (&a)[NBLK], double& yk, double& wk) __attribute__((always_inline)) {
            const int roff = (4 * s + fk) * ES;
#pragma unroll
            for (int b = 0; b < NBLK; ++b) a[b] = (double)*(const __attribute__((address_space(3))) T*)(base + fi * GS + b * HR * ES + roff);
            yk = (double)*(const __attribute__((address_space(3))) T*)(base + MD::Y_OFF + roff);
            // (FUSE: the weight image is made in the kernel, in f64 -- and FUSE is the f64 kernel's)
            if constexpr (WT) wk = FUSE ? *(const __attribute__((address_space(3))) double*)(base + MD::W_OFF + roff)
                                        : (double)*(const __attribute__((address_space(3))) T*)(base + MD::W_OFF + roff);
        };
        // operands of step s + 1 are fetched from LDS before step s multiplies (one wave per SIMD: nobody else hides the round trip)
        double a[NBLK], yk, wk = 1.0;
        fetch(0, a, yk, wk);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            double an[NBLK], ykn = 0.0, wkn = 1.0;
#pragma unroll
            for (int b = 0; b < NBLK; ++b) an[b] = 0.0;
            if (s + 1 < NS) fetch(s + 1, an, ykn, wkn);
            if (FUSE == 0 && next) {
#pragma unroll
                for (int cc = 0; cc < IPS; ++cc)
                    if (s * IPS + cc <= 16) issue_one(s * IPS + cc, buf ^ 1, next_row0);
            }
            // weighted: the A operand carries w x, the B operand x -- sum_k (w_k x_ki) x_kj
            double aw[NBLK];
#pragma unroll
            for (int b = 0; b < NBLK; ++b) aw[b] = WT ? a[b] * wk : a[b];
            // (FUSE: the record is read as a MEAT block -- X'WX, its column sums and sum w; X'y, y'y and sum y are not formed: f64 vector
            //  instructions take the SIMD's FP64 unit from the matrix instructions)
            const double ywk = FUSE ? 0.0 : (WT ? yk * wk : yk);
            int q = 0;
#pragma unroll
            for (int I = 0; I < NBLK; ++I)
#pragma unroll
                for (int J = I; J < NBLK; ++J) {
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[I], a[J], acc[q], 0, 0, 0);
                    ++q;
                }
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                if constexpr (FUSE == 0) xy[b] = fma(aw[b], yk, xy[b]);
                cs[b] += aw[b];
            }
            if constexpr (FUSE == 0) {
                yy = fma(ywk, yk, yy);
                ys += ywk;
            }
            if constexpr (WT) sw += wk;
#pragma unroll
            for (int b = 0; b < NBLK; ++b) a[b] = an[b];
            yk = ykn;
            wk = wkn;
        }
    };
    if (h0 < h1) {
        issue(0, h0 * HR);
        for (int64_t h = h0; h < h1; ++h) {
            const int buf = (int)((h - h0) & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // half-tile h has landed (the compiler does not order LDS reads behind it)
            __builtin_amdgcn_wave_barrier();
            if constexpr (FUSE != 0) {
                produce(buf, HR, h + 1 < h1, (h + 1) * HR);
                PDS_WAVE_LDS_SYNC();
            }
            consume(buf, h + 1 < h1, (h + 1) * HR);  // (the other image was consumed one iteration ago: free for the next half-tile)
            PDS_WAVE_LDS_SYNC();
        }
    }
    if (tail > 0) {
        load_tail(0, nh * HR, tail);
        PDS_WAVE_LDS_SYNC();
        if constexpr (FUSE != 0) {
            produce(0, tail, false, 0);
            PDS_WAVE_LDS_SYNC();
        }
        consume(0, false, 0);
    }
    // ---- the wave's partial record
    double* rec = partials + (size_t)wave * MD::REC;
#pragma unroll
    for (int q = 0; q < NPAIR; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) rec[(q * 4 + r) * 64 + lane] = acc[q][r];
    double* v = rec + NPAIR * 256;
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        v[b * 64 + lane] = xy[b];
        v[(NBLK + b) * 64 + lane] = cs[b];
    }
    v[2 * NBLK * 64 + lane] = yy;
    v[2 * NBLK * 64 + 64 + lane] = ys;
    v[2 * NBLK * 64 + 128 + lane] = sw;
    v[2 * NBLK * 64 + 192 + lane] = sse;
#undef PDS_MID_LDSD
}

// per-wave records -> one record: entry idx summed over the waves in a fixed order (four interleaved partial sums, then their sum)
__global__ __launch_bounds__(256) void moments_mid_reduce_kernel(const double* __restrict__ partials, int nwaves, int rec, double* __restrict__ out) {
    __shared__ double part[4][64];
    const int i = threadIdx.x & 63, g = threadIdx.x >> 6, idx = blockIdx.x * 64 + i;
    double s = 0.0;
    if (idx < rec)
        for (int w = g; w < nwaves; w += 4) s += partials[(size_t)w * rec + idx];
    part[g][i] = s;
    __syncthreads();
    if (g == 0 && idx < rec) out[idx] = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
}

// one thread per entry (i <= j) of the (p+2)^2 moment matrix over [x_0 .. x_{p-1}, 1, y]: fixed-order sum over the waves
template <int NBLK, typename T = double>
__global__ __launch_bounds__(256) void moments_mid_finalize_kernel(const double* __restrict__ partials, int nwaves, int p, int64_t n, int weighted,
                                                                   T* __restrict__ out, double* __restrict__ sums /* nullable: [sum e^2, 0] */) {
    using MD = MidDims<NBLK>;
    const int q = p + 2;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0 && sums) {
        double t = 0.0;
        for (int w = 0; w < nwaves; ++w) {
            const double* v = partials + (size_t)w * MD::REC + MD::NPAIR * 256 + 2 * NBLK * 64 + 192;
            for (int l = 0; l < 64; l += 4) t += (v[l] + v[l + 1]) + (v[l + 2] + v[l + 3]);
        }
        sums[0] = t;
        sums[1] = 0.0;
    }
    if (e >= q * q) return;
    int i = e % q, j = e / q;
    if (i > j) {
        const int t = i;
        i = j;
        j = t;
    }
    double s = 0.0;
    if (j < p) {  // X'X: tile (I, J), D[row = ii][col = jj] sits in register ii / 4 of lane jj + 16 (ii % 4)
        const int I = i / 16, J = j / 16, ii = i % 16, jj = j % 16;
        int pair = 0;
        for (int a = 0; a < I; ++a) pair += NBLK - a;
        pair += J - I;
        const int idx = (pair * 4 + ii / 4) * 64 + jj + 16 * (ii % 4);
        for (int w = 0; w < nwaves; ++w) s += partials[(size_t)w * MD::REC + idx];
    } else if (i < p) {  // column sums (j == p) or X'y (j == p + 1): the four row slots of feature i
        const int b = i / 16, ii = i % 16;
        const int base = MD::NPAIR * 256 + ((j == p ? NBLK : 0) + b) * 64;
        for (int w = 0; w < nwaves; ++w) {
            const double* v = partials + (size_t)w * MD::REC + base;
            s += (v[ii] + v[ii + 16]) + (v[ii + 32] + v[ii + 48]);
        }
    } else if (i == p && j == p) {
        if (weighted) {
            const int base = MD::NPAIR * 256 + 2 * NBLK * 64 + 128;
            for (int w = 0; w < nwaves; ++w) {
                const double* v = partials + (size_t)w * MD::REC + base;
                s += (v[0] + v[16]) + (v[32] + v[48]);
            }
        } else {
            s = (double)n;
        }
    } else {  // sum y (i == p) or y'y (i == p + 1): lanes 0, 16, 32, 48 hold the four row slots
        const int base = MD::NPAIR * 256 + 2 * NBLK * 64 + (i == p ? 64 : 0);
        for (int w = 0; w < nwaves; ++w) {
            const double* v = partials + (size_t)w * MD::REC + base;
            s += (v[0] + v[16]) + (v[32] + v[48]);
        }
    }
    out[e] = (T)s;
}

template <int NBLK, bool WEIGHTED, int FUSE, typename T = double>
int launch_mid(pds_ctx* ctx, const DeviceCols<T>& dc, int p, int64_t n, T* d_moments, int bias = 0, const double* d_beta = nullptr,
               const double* d_lop = nullptr, int hc = 0, double* d_sums = nullptr) {
    using MD = MidDims<NBLK, (int)sizeof(T)>;
    using MS = MidShared<NBLK, FUSE>;
    const int nwaves = ctx->num_cus * kMidWavesPerCu;
    double* partials = reinterpret_cast<double*>(ws_take(ctx, (size_t)(nwaves + 1) * MD::REC * sizeof(double)));
    if (!partials) return fail(PDS_ERR_HIP, "workspace allocation failed");
    auto kern = moments_mid_kernel<NBLK, WEIGHTED, FUSE, T>;
    constexpr int lds = MS::BYTES + MS::WPB * MD::LDS_BYTES;
    static_assert(lds <= 160 * 1024, "one workgroup per CU at most");
    if (lds > 64 * 1024) PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    KernelTimer timer(ctx, FUSE ? kKindPass2 : kKindMoments);
    hipLaunchKernelGGL(kern, dim3(nwaves / MS::WPB), dim3(64 * MS::WPB), lds, ctx->stream, dc.d_ptrs, p, n, partials, bias, d_beta, d_lop, hc);
    const int q = p + 2;
    double* reduced = partials + (size_t)nwaves * MD::REC;
    hipLaunchKernelGGL(moments_mid_reduce_kernel, dim3((MD::REC + 63) / 64), dim3(256), 0, ctx->stream, (const double*)partials, nwaves, MD::REC, reduced);
    hipLaunchKernelGGL((moments_mid_finalize_kernel<NBLK, T>), dim3((q * q + 255) / 256), dim3(256), 0, ctx->stream, (const double*)reduced, 1, p, n,
                       (WEIGHTED || FUSE) ? 1 : 0, d_moments, d_sums);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}


}  // namespace

size_t moments_mid_workspace(int num_cus) { return (size_t)(num_cus * kMidWavesPerCu + 1) * MidDims<4>::REC * sizeof(double) + 4096; }

// 17 .. 64 features (weights: table entry p + 1): d_moments = (p+2)^2 column-major over [x_0 .. x_{p-1}, 1, y].  f32 frames: the
// values are widened on their way out of LDS -- exact f32 products, f64 sums on the f64 matrix instruction -- and the record is
// rounded to f32 once at the end.
template <typename T>
int launch_moments_mid(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, bool weighted, T* d_moments) {
    if (n_feat <= 32)
        return weighted ? launch_mid<2, true, 0, T>(ctx, dc, n_feat, n_rows, d_moments) : launch_mid<2, false, 0, T>(ctx, dc, n_feat, n_rows, d_moments);
    if (n_feat <= 64)
        return weighted ? launch_mid<4, true, 0, T>(ctx, dc, n_feat, n_rows, d_moments) : launch_mid<4, false, 0, T>(ctx, dc, n_feat, n_rows, d_moments);
    return fail(PDS_ERR_UNSUPPORTED, "moments_mid: up to 64 features");
}
template int launch_moments_mid<double>(pds_ctx*, const DeviceCols<double>&, int, int64_t, bool, double*);
template int launch_moments_mid<float>(pds_ctx*, const DeviceCols<float>&, int, int64_t, bool, float*);

// The second pass of an UNWEIGHTED report with robust errors, 17 .. 64 f64 features, as one stream: d_sums = [sum e^2, 0] and
// d_meat = the (p+2)^2 moment layout of X' diag(s) X, s = e^2 (hc_mode 1) or e^2 / (1 - h)^(hc_mode - 1) (2, 3; d_inv = (X'X)^-1).
// PDS_ERR_UNSUPPORTED (nothing done): the inverse has no Cholesky factor -- the caller keeps the three-kernel form.
int launch_report_mid(pds_ctx* ctx, const DeviceCols<double>& dc, int n_feat, int bias, int64_t n_rows, const double* d_beta, const double* d_inv,
                      int hc_mode, double* d_sums, double* d_meat) {
    if (n_feat < 17 || n_feat > 64 || hc_mode < 1) return PDS_ERR_UNSUPPORTED;
    if (hc_mode == 1) {
        if (n_feat <= 32) return launch_mid<2, false, 1>(ctx, dc, n_feat, n_rows, d_meat, bias, d_beta, nullptr, 1, d_sums);
        return launch_mid<4, false, 1>(ctx, dc, n_feat, n_rows, d_meat, bias, d_beta, nullptr, 1, d_sums);
    }
    const double* d_lop = nullptr;
    if (int rc = leverage_operand(ctx, d_inv, n_feat, bias, &d_lop)) return rc;
    if (n_feat <= 32) return launch_mid<2, false, 2>(ctx, dc, n_feat, n_rows, d_meat, bias, d_beta, d_lop, hc_mode, d_sums);
    return launch_mid<4, false, 2>(ctx, dc, n_feat, n_rows, d_meat, bias, d_beta, d_lop, hc_mode, d_sums);
}

}  // namespace pds
