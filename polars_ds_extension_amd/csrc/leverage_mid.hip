// leverage_mid.hip -- HC2 / HC3 leverages h_r = z_r' (X'X)^-1 z_r for 17 .. 64 f64 features on the matrix cores.
//
// pl_lin_reg_report's HC2 / HC3 (linear_regression.rs:880-909) scale every squared residual by 1 / (1 - h_r)^k.  Beyond 16 features
// the leverages were a per-row O(p'^2) product on the vector ALU (leverage_scale_wide_regs_kernel, pass2.hip: 13.4 ms of the 22 ms
// of the report at 2e7 x 64).  They are an n x p' x p' product, which belongs on the matrix cores:
//     (X'X)^-1 = L L'  (Cholesky, p' x p', on the host: 65^3 / 3 flops)      h_r = || L' z_r ||^2
// so only the lower triangle of L takes part and the row's quadratic form becomes a sum of squares.  The kernel streams the frame
// exactly like moments_mid.hip (wave-private LDS images, 1 KiB asynchronous global -> LDS loads of NBLK columns each, one wave per
// SIMD); per 16 rows and 16 outputs a it runs the k-steps b >= a of v_mfma_f64_16x16x4_f64 with A = the rows' features (lane = (row,
// feature)) and B = a 4 x 16 block of L read from an operand-ordered copy in LDS (shared by the workgroup's four waves), squares the
// tile and sums it over a with DPP row reductions.  The intercept is a constant row of L: it initialises the accumulator.
// Matrix work: 40 instructions per 16 rows at 64 features (1.9 ms of pipe per 2e7 rows), 12 at 32.
#include "common.hpp"
#include "moments_dev.hpp"

#include <cmath>
#include <vector>

namespace pds {

namespace {

template <int NBLK>
struct LevDims {
    static constexpr int HR = 128 / NBLK;       // rows per half-tile
    static constexpr int GL = 64 / NBLK;        // lanes per lane group of a load instruction
    static constexpr int GS = 1024 + 16;        // bytes between the images of load instructions i and i + 1
    static constexpr int IMG = 16 * GS;         // one image (features only)
    static constexpr int KS = 4 * NBLK;         // k-steps (4 features each)
    static constexpr int NLB = 4 * (NBLK * (NBLK + 1) / 2);  // L operand blocks: k-steps >= 4 ablk for every ablk
    static constexpr int L_BYTES = NLB * 512 + 16 * NBLK * 8 + 16;  // blocks | intercept row | constant
    static constexpr int LDS_BYTES = L_BYTES + 4 * 2 * IMG;
};

// index of the operand block (ablk, kstep), kstep >= 4 ablk
template <int NBLK>
__host__ __device__ constexpr int lev_block(int ablk, int kstep) {
    int base = 0;
    for (int a = 0; a < ablk; ++a) base += 4 * NBLK - 4 * a;
    return base + (kstep - 4 * ablk);
}

template <int NBLK>
__global__ __launch_bounds__(256) void leverage_mid_kernel(const double* const* __restrict__ cols, int p, int bias, int64_t n,
                                                           const double* __restrict__ lop /*L_BYTES / 8 doubles*/, int hc,
                                                           double* __restrict__ s_rows) {
    using LD = LevDims<NBLK>;
    constexpr int HR = LD::HR, GS = LD::GS, KS = LD::KS;
    extern __shared__ __attribute__((aligned(16))) char lev_lds[];
    typedef __attribute__((address_space(3))) char* lds_c;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    typedef double lev_d2 __attribute__((ext_vector_type(2)));
    lds_c sm = (lds_c)lev_lds;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- the workgroup's copy of L (operand order), then this wave's two images, zeroed once (padding columns stay zero)
    for (int i = tid; i < LD::L_BYTES / 8; i += 256) *(__attribute__((address_space(3))) double*)(sm + i * 8) = lop[i];
    lds_c img = sm + LD::L_BYTES + wv * 2 * LD::IMG;
    for (int i = lane * 16; i < 2 * LD::IMG; i += 64 * 16) *(__attribute__((address_space(3))) lev_d2*)(img + i) = lev_d2{0.0, 0.0};
    __syncthreads();
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    const int64_t nh = n / HR;
    const int64_t h0 = nh * wave / nwaves, h1 = nh * (wave + 1) / nwaves;
    const int tail = (wave == nwaves - 1) ? (int)(n - nh * HR) : 0;
    const int g = lane / LD::GL, piece = lane % LD::GL;
    const double* cbase[16];
    unsigned valid = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = 16 * g + i;
        cbase[i] = cols[c < p ? c : 0] + 2 * piece;
        if (c < p) valid |= 1u << i;
    }
    auto issue_one = [&](int i, int buf, int64_t row0) __attribute__((always_inline)) {
        if ((valid >> i) & 1u)
            __builtin_amdgcn_global_load_lds((glb_ptr)(as_global(cbase[i]) + row0), (lds_ptr)(img + buf * LD::IMG + i * GS), 16, 0, 0);
    };
    auto load_tail = [&](int buf, int64_t row0, int rows) __attribute__((always_inline)) {
        for (int c = 0; c < p; ++c) {
            const int off = (c % 16) * GS + (c / 16) * HR * 8;
            const gptr<double> col = as_global(cols[c]);
            for (int r = lane; r < HR; r += 64)
                *(__attribute__((address_space(3))) double*)(img + buf * LD::IMG + off + r * 8) = r < rows ? col[row0 + r] : 0.0;
        }
    };
    const int fi = lane & 15, fk = lane >> 4;
    const lds_c lblk = sm;
    const lds_c lbias = sm + LD::NLB * 512;
    const double c0 = bias ? *(const __attribute__((address_space(3))) double*)(sm + LD::NLB * 512 + 16 * NBLK * 8) : 0.0;
    // one half-tile: `rows` valid rows at row0; the next half-tile's loads go out between the row groups
    auto consume = [&](int buf, int64_t row0, int rows, bool next, int64_t next_row0) __attribute__((always_inline)) {
        const lds_c base = img + buf * LD::IMG;
        constexpr int RG = HR / 16, LPG = (16 + RG - 1) / RG;  // row groups; load instructions issued per row group
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            if (next) {
#pragma unroll
                for (int cc = 0; cc < LPG; ++cc)
                    if (rg * LPG + cc < 16) issue_one(rg * LPG + cc, buf ^ 1, next_row0);
            }
            double hs[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ablk = 0; ablk < NBLK; ++ablk) {
                const double init = bias ? *(const __attribute__((address_space(3))) double*)(lbias + (16 * ablk + fi) * 8) : 0.0;
                d4 acc = d4{init, init, init, init};
#pragma unroll
                for (int ks = 4 * ablk; ks < KS; ++ks) {
                    const int f = 4 * ks + fk;  // this lane's feature of the step (A operand: lane = (row i, feature k))
                    const double a = *(const __attribute__((address_space(3))) double*)(base + (f & 15) * GS + (f >> 4) * HR * 8 + (16 * rg + fi) * 8);
                    const double b = *(const __attribute__((address_space(3))) double*)(lblk + lev_block<NBLK>(ablk, ks) * 512 + lane * 8);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) hs[r] = fma(acc[r], acc[r], hs[r]);
            }
            // sum over the 16 outputs a of the lane row: D has col = lane & 15 = a, row = (lane >> 4) + 4 r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = hs[r];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                hs[r] = v + c0;
            }
            if (fi == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = 16 * rg + fk + 4 * r;
                    if (lr < rows) {
                        const double om = 1.0 - hs[r];
                        const double sc = (hc == 2) ? 1.0 / om : 1.0 / (om * om);
                        s_rows[row0 + lr] *= sc;
                    }
                }
            }
        }
    };
    if (h0 < h1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) issue_one(i, 0, h0 * HR);
        for (int64_t h = h0; h < h1; ++h) {
            const int buf = (int)((h - h0) & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // half-tile h has landed (incl. this wave's s_rows stores: cheap next to 40 matrix steps)
            __builtin_amdgcn_wave_barrier();
            consume(buf, h * HR, HR, h + 1 < h1, (h + 1) * HR);
            PDS_WAVE_LDS_SYNC();
        }
    }
    if (tail > 0) {
        load_tail(0, nh * HR, tail);
        PDS_WAVE_LDS_SYNC();
        consume(0, nh * HR, tail, false, 0);
    }
}

template <int NBLK>
void fill_lev_operand(const std::vector<double>& L, int p, int pp, int bias, std::vector<double>& lop) {
    using LD = LevDims<NBLK>;
    // operand-ordered copy of L: block (ablk, kstep): lane (k = lane >> 4, j = lane & 15) holds L[4 kstep + k][16 ablk + j]
    lop.assign(LD::L_BYTES / 8, 0.0);
    for (int ablk = 0; ablk < NBLK; ++ablk)
        for (int ks = 4 * ablk; ks < 4 * NBLK; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int b = 4 * ks + (lane >> 4), a = 16 * ablk + (lane & 15);
                if (b < p && a < p && b >= a) lop[(size_t)lev_block<NBLK>(ablk, ks) * 64 + lane] = L[b + (size_t)a * pp];
            }
    if (bias) {
        for (int a = 0; a < p; ++a) lop[(size_t)LD::NLB * 64 + a] = L[p + (size_t)a * pp];  // row p of L: the intercept's contribution
        lop[(size_t)LD::NLB * 64 + 16 * NBLK] = L[p + (size_t)p * pp] * L[p + (size_t)p * pp];
    }
}

template <int NBLK>
int launch_lev(pds_ctx* ctx, const DeviceCols<double>& dc, int p, int bias, int64_t n, const double* d_lop, int hc, double* s_rows) {
    using LD = LevDims<NBLK>;
    auto kern = leverage_mid_kernel<NBLK>;
    PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LD::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(ctx->num_cus), dim3(256), LD::LDS_BYTES, ctx->stream, dc.d_ptrs, p, bias, n, d_lop, hc, s_rows);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

}  // namespace

// The operand-ordered Cholesky factor of d_inv (p' x p' column-major on the device; inv = L L') in the workspace of ctx:
//   [ blocks (ablk, kstep >= 4 ablk) of 64 doubles | row p of L (16 NBLK doubles) | L[p][p]^2, 0 ],  NBLK = 2 (p <= 32) or 4.
// Returns PDS_ERR_UNSUPPORTED (nothing done) when the inverse has no Cholesky factor in f64.  Synchronises the stream.
int leverage_operand(pds_ctx* ctx, const double* d_inv, int n_feat, int bias, const double** d_lop) {
    if (n_feat < 17 || n_feat > 64) return PDS_ERR_UNSUPPORTED;
    const int pp = n_feat + (bias ? 1 : 0);
    std::vector<double> A((size_t)pp * pp), L((size_t)pp * pp, 0.0);
    PDS_HIP_CHECK(hipMemcpyAsync(A.data(), d_inv, A.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < pp; ++j) {  // A = L L', column by column
        double d = A[j + (size_t)j * pp];
        for (int k = 0; k < j; ++k) d -= L[j + (size_t)k * pp] * L[j + (size_t)k * pp];
        if (!(d > 0.0) || !std::isfinite(d)) return PDS_ERR_UNSUPPORTED;
        const double ljj = std::sqrt(d);
        L[j + (size_t)j * pp] = ljj;
        for (int i = j + 1; i < pp; ++i) {
            double s = 0.5 * (A[i + (size_t)j * pp] + A[j + (size_t)i * pp]);
            for (int k = 0; k < j; ++k) s -= L[i + (size_t)k * pp] * L[j + (size_t)k * pp];
            L[i + (size_t)j * pp] = s / ljj;
        }
    }
    std::vector<double> lop;
    if (n_feat <= 32) fill_lev_operand<2>(L, n_feat, pp, bias, lop);
    else fill_lev_operand<4>(L, n_feat, pp, bias, lop);
    double* d = reinterpret_cast<double*>(ws_take(ctx, lop.size() * sizeof(double)));
    if (!d) return fail(PDS_ERR_HIP, "workspace allocation failed");
    PDS_HIP_CHECK(hipMemcpyAsync(d, lop.data(), lop.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (lop: source of the copy)
    *d_lop = d;
    return PDS_OK;
}

// s_rows[r] *= 1 / (1 - h_r)^(hc - 1), h_r = z_r' inv z_r, z = [x_0 .. x_{p-1}, (1)]; d_inv: p' x p' column-major on the device.
// Returns PDS_OK, or PDS_ERR_UNSUPPORTED (nothing done) when the inverse has no Cholesky factor in f64 -- the caller then keeps
// the vector-ALU form.
int launch_leverage_mid(pds_ctx* ctx, const DeviceCols<double>& dc, int n_feat, int bias, int64_t n_rows, const double* d_inv, int hc_mode,
                        double* d_s_rows) {
    const double* d_lop = nullptr;
    if (int rc = leverage_operand(ctx, d_inv, n_feat, bias, &d_lop)) return rc;
    KernelTimer timer(ctx, kKindPass2);
    if (n_feat <= 32) return launch_lev<2>(ctx, dc, n_feat, bias, n_rows, d_lop, hc_mode, d_s_rows);
    return launch_lev<4>(ctx, dc, n_feat, bias, n_rows, d_lop, hc_mode, d_s_rows);
}

}  // namespace pds
