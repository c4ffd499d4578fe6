// rolling_seg.hip -- EXPERIMENT (compile-only so far; not linked into the library, never run on a GPU yet).
//
// A different decomposition of the rolling fit: today lane = row of a 64-row step and the NV = p'(p'+1)/2
// + p' + 1 running moments make a round trip through LDS per step (increments out, a 32-long scan by two lanes per
// moment, window sums back): per step and wave 12 400 clk, of which the solve itself is a small part.
// Here lane = K CONSECUTIVE rows of a 64 K-row stage and everything stays in registers:
//   pass 1   D_l   = sum over the lane's K rows of (m(r) - m(r - w))                  2 NG' FMAs per row
//   scan     P_l   = carry + exclusive prefix of D over the lanes (DPP, 6 steps)      18 instructions per moment PER STAGE
//   pass 2   S = P_l; per row: S += m(r) - m(r - w); solve S beta = c; pred; store    2 NG' FMAs + the solve per row
// with m(r) the moment vector of row r.  No LDS, no barrier, the scan amortised over K rows.  Loads: the lane's K rows of
// a column are K*8 contiguous bytes and the lanes are contiguous, so a stage reads 64 K rows of every column as whole lines.
// Tiles (kTile rows) are anchored exactly as in rolling.hip: the window in front of the tile is summed cooperatively.
//
// Static numbers (hipcc 7.2, -O3, gfx950; `python tools/experiments/isa_mix.py /tmp/rolling_seg.s`):
//   <p' = 8, K = 4>  470 VGPRs (one wave per SIMD), 8 spills to AGPRs; a 256-row stage = ~3 900 instructions: 2 060 f64
//                    arithmetic, 540 DPP moves, ~700 register moves  ->  975 per 64 rows ~ 3 900 clk per wave
//   <p' = 8, K = 2>  348 VGPRs, a 128-row stage = ~2 440 instructions -> 1 220 per 64 rows (the scan amortises over fewer rows)
//   <p' = 5, K = 4>  298 VGPRs, a stage = ~1 630 instructions -> 410 per 64 rows
// Today's kernel: 12 400 clk per 64-row step and wave with two waves per SIMD = 6 200 clk per 64 rows and SIMD.  If the loads
// hide behind one wave per SIMD (prefetch a stage ahead: registers are there at K = 4 only after the moves are gone) this
// is 1.6x at p' = 8; the floor is the 2 060 arithmetic instructions (2 060 clk per 64 rows: 3x).  Unmeasured.
// What the static numbers are for: VGPRs / scratch decide whether K = 4 (data of both passes held in registers) fits;
// the VALU count per stage bounds the step from below.   hipcc -O3 --offload-arch=gfx950 -S -o - tools/experiments/rolling_seg.hip
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pds_experiment {

template <typename T>
using gptr = const __attribute__((address_space(1))) T*;
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

constexpr int kTile = 4096;

// x_i += x_{i - s} inside 16-lane rows (s = 1, 2, 4, 8), then across rows: an inclusive scan over the 64 lanes
// (ROW_MASK 0xf with bound_ctrl: every lane is written, lanes whose source falls outside the row read 0 -- no `old` register
//  to initialise; the two cross-row steps write only some rows and need the zero)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double x) {
    constexpr bool kAll = ROW_MASK == 0xf;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, kAll);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, kAll);
    return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_inclusive_scan(double x) {
    x = dpp_add<0x111, 0xf>(x);  // row_shr:1
    x = dpp_add<0x112, 0xf>(x);  // row_shr:2
    x = dpp_add<0x114, 0xf>(x);  // row_shr:4
    x = dpp_add<0x118, 0xf>(x);  // row_shr:8
    x = dpp_add<0x142, 0xa>(x);  // row_bcast:15 -> rows 1, 3
    x = dpp_add<0x143, 0xc>(x);  // row_bcast:31 -> rows 2, 3
    return x;
}
__device__ __forceinline__ double lane63(double x) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
__device__ __forceinline__ double uniform(double x) {  // the value is the same in every lane: keep it in a scalar register pair
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

template <int PP>
struct Row {
    double z[PP], y;
    bool ok;
};

template <int PP, int K>
__device__ __forceinline__ void load_rows(const double* const* __restrict__ cols, int64_t r0, int64_t n, Row<PP> (&rows)[K]) {
    static_assert(K % 2 == 0, "two rows per 16-byte load");
#pragma unroll
    for (int c = 0; c <= PP; ++c) {
        const double* col = cols[c];
#pragma unroll
        for (int i = 0; i < K; i += 2) {
            const int64_t r = r0 + i;
            d2u v = {0.0, 0.0};
            if (r >= 0 && r + 1 < n) v = *(const __attribute__((address_space(1))) d2u*)(col + r);
            else if (r >= 0 && r < n) v.x = ((gptr<double>)col)[r];
            if (c < PP) {
                rows[i].z[c] = v.x;
                rows[i + 1].z[c] = v.y;
            } else {
                rows[i].y = v.x;
                rows[i + 1].y = v.y;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
        bool fin = isfinite(rows[i].y) && (r0 + i >= 0) && (r0 + i < n);
#pragma unroll
        for (int a = 0; a < PP; ++a) fin = fin && isfinite(rows[i].z[a]);
        rows[i].ok = fin;
        if (!fin) {
#pragma unroll
            for (int a = 0; a < PP; ++a) rows[i].z[a] = 0.0;
            rows[i].y = 0.0;
        }
    }
}

// S += sign * m(row): Gram upper triangle, X'y, count
template <int PP, int NV>
__device__ __forceinline__ void accumulate(double (&S)[NV], const Row<PP>& r, double sign) {
    int v = 0;
#pragma unroll
    for (int a = 0; a < PP; ++a) {
        const double sa = sign * r.z[a];
#pragma unroll
        for (int b = a; b < PP; ++b) S[v++] = fma(sa, r.z[b], S[v]);
    }
#pragma unroll
    for (int a = 0; a < PP; ++a) S[v++] = fma(sign * r.z[a], r.y, S[v]);
    S[v] += r.ok ? sign : 0.0;
}

template <int PP, int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rolling_seg_kernel(
    const double* const* __restrict__ cols, int64_t n, int64_t w, double lambda, int64_t min_size, double* __restrict__ coeffs,
    double* __restrict__ pred, uint8_t* __restrict__ valid) {
    constexpr int NG = PP * (PP + 1) / 2, NV = NG + PP + 1, STAGE = 64 * K;
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    const int64_t ntiles = (n + kTile - 1) / kTile;
    for (int64_t t = wid; t < ntiles; t += nw) {
        const int64_t t0 = t * kTile, t1 = (t0 + kTile < n) ? t0 + kTile : n;
        // ---- anchor: the window sum at row t0 - 1 (rows t0 - w ... t0 - 1), lanes stride over the rows, then one wave sum
        double carry[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) carry[v] = 0.0;
        for (int64_t r = t0 - w + 2 * lane; r < t0; r += 128) {
            Row<PP> two[2];
            load_rows<PP, 2>(cols, r, (r + 1 < t0) ? n : (r + 1 <= n ? r + 1 : n), two);  // (the pair may straddle t0)
            accumulate<PP, NV>(carry, two[0], 1.0);
            accumulate<PP, NV>(carry, two[1], 1.0);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) carry[v] = lane63(wave_inclusive_scan(carry[v]));
        for (int64_t base = t0; base < t1; base += STAGE) {
            const int64_t r0 = base + (int64_t)K * lane;
            Row<PP> rn[K], ro[K];
            load_rows<PP, K>(cols, r0, t1, rn);
            load_rows<PP, K>(cols, r0 - w, n, ro);
#pragma unroll
            for (int i = 0; i < K; ++i)  // rows behind the tile end contribute nothing; their old rows neither
                if (r0 + i >= t1) {
                    ro[i].ok = false;
#pragma unroll
                    for (int a = 0; a < PP; ++a) ro[i].z[a] = 0.0;
                    ro[i].y = 0.0;
                }
            // ---- pass 1: the lane's own increments
            double S[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) S[v] = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                accumulate<PP, NV>(S, rn[i], 1.0);
                accumulate<PP, NV>(S, ro[i], -1.0);
            }
            // ---- scan: start state of lane l = carry + increments of the lanes in front; the wave total is the next carry
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const double incl = wave_inclusive_scan(S[v]);
                S[v] = carry[v] + (incl - S[v]);
                carry[v] = uniform(carry[v] + lane63(incl));
            }
            // ---- pass 2: row by row
#pragma unroll
            for (int i = 0; i < K; ++i) {
                accumulate<PP, NV>(S, rn[i], 1.0);
                accumulate<PP, NV>(S, ro[i], -1.0);
                const int64_t r = r0 + i;
                // L D L' of (G + lambda I) in a copy; idx(a,b), a <= b -> a*PP - a(a-1)/2 + (b-a)
                double g[NG], c[PP], rd[PP];
#pragma unroll
                for (int a = 0; a < PP; ++a) c[a] = S[NG + a];
#define GI(a, b) g[(a) * PP - ((a) * ((a)-1)) / 2 + ((b) - (a))]
#define SI(a, b) S[(a) * PP - ((a) * ((a)-1)) / 2 + ((b) - (a))]
                bool okc = true;
                {   // step 0 reads the running sums and writes the work copy: no register copy of the 36 values
                    const double d = SI(0, 0) + lambda;
                    okc = d > 0.0;
                    double x = __builtin_amdgcn_rcp(d);
                    x = x * fma(-d, x, 2.0);
                    x = x * fma(-d, x, 2.0);
                    rd[0] = x;
#pragma unroll
                    for (int a = 1; a < PP; ++a) {
                        const double tka = SI(0, a) * x;
#pragma unroll
                        for (int b = a; b < PP; ++b) GI(a, b) = fma(-tka, SI(0, b), SI(a, b) + ((a == b) ? lambda : 0.0));
                        GI(0, a) = tka;
                    }
                }
#pragma unroll
                for (int k = 1; k < PP; ++k) {
                    const double d = GI(k, k);
                    okc = okc && (d > 0.0);
                    double x = __builtin_amdgcn_rcp(d);
                    x = x * fma(-d, x, 2.0);
                    x = x * fma(-d, x, 2.0);
                    rd[k] = x;
#pragma unroll
                    for (int a = k + 1; a < PP; ++a) {
                        const double tka = GI(k, a) * x;  // l_ak
#pragma unroll
                        for (int b = a; b < PP; ++b) GI(a, b) = fma(-tka, GI(k, b), GI(a, b));
                        GI(k, a) = tka;
                    }
                }
#undef SI
#pragma unroll
                for (int a = 1; a < PP; ++a)
#pragma unroll
                    for (int k = 0; k < a; ++k) c[a] = fma(-GI(k, a), c[k], c[a]);
#pragma unroll
                for (int a = 0; a < PP; ++a) c[a] *= rd[a];
#pragma unroll
                for (int a = PP - 2; a >= 0; --a)
#pragma unroll
                    for (int k = a + 1; k < PP; ++k) c[a] = fma(-GI(a, k), c[k], c[a]);
#undef GI
                if (r < t1) {
                    const double nanv = __builtin_nan("");
                    bool v_ok = r >= w - 1;
                    if (min_size > 0) v_ok = v_ok && (S[NV - 1] >= (double)min_size);
                    double pr = 0.0;
#pragma unroll
                    for (int a = 0; a < PP; ++a) pr = fma(rn[i].z[a], c[a], pr);
                    if (!rn[i].ok) pr = nanv;
                    double* out = coeffs + r * PP;
#pragma unroll
                    for (int a = 0; a < PP; ++a) out[a] = (v_ok && okc) ? c[a] : nanv;
                    pred[r] = (v_ok && okc) ? pr : nanv;
                    valid[r] = v_ok ? 1 : 0;
                }
            }
        }
    }
}

template __global__ void rolling_seg_kernel<8, 4>(const double* const*, int64_t, int64_t, double, int64_t, double*, double*, uint8_t*);
template __global__ void rolling_seg_kernel<8, 2>(const double* const*, int64_t, int64_t, double, int64_t, double*, double*, uint8_t*);
template __global__ void rolling_seg_kernel<5, 4>(const double* const*, int64_t, int64_t, double, int64_t, double*, double*, uint8_t*);

}  // namespace pds_experiment
