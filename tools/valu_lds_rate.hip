// tools/valu_lds_rate.hip -- (1) v_fma_f64 with ONE / TWO / THREE vector-register sources per wave (the in-register solves are
// all three-VGPR-source FMAs); (2) LDS f64 update rates: ds_add_f64 (hardware atomic) against ds_read_b64 + v_add_f64 + ds_write_b64,
// random and conflict-free addresses.  One wave per SIMD unless said otherwise.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu2 tools/valu_lds_rate.hip && /tmp/valu2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE, int NACC>
__global__ void fma3_kernel(double* out, const double* in, int iters, unsigned long long* clk) {
    double acc[NACC], x[NACC], y[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        acc[i] = in[threadIdx.x + i];
        x[i] = in[threadIdx.x + 64 + i] * 1e-9 + 1.0;
        y[i] = in[threadIdx.x + 128 + i] * 1e-9;
    }
    const double s = in[0] * 1e-9 + 1.0, t = in[1] * 1e-9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (MODE == 1) acc[i] = __builtin_fma(acc[i], s, t);          // one VGPR source (+ two uniform)
                if (MODE == 2) acc[i] = __builtin_fma(x[i], s, acc[i]);       // two VGPR sources
                if (MODE == 3) acc[i] = __builtin_fma(x[i], y[(i + r) % NACC], acc[i]);  // three VGPR sources
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double sum = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) sum += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// MODE 0: ds_add_f64; 1: read + add + write.  PATTERN 0: lane-consecutive addresses, 1: pseudo-random
template <int MODE, int PATTERN>
__global__ void lds_kernel(double* out, int iters, unsigned long long* clk) {
    __shared__ double buf[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = 0.0;
    __syncthreads();
    typedef __attribute__((address_space(3))) double* lds_d;
    unsigned a = threadIdx.x * 2654435761u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            unsigned idx;
            if (PATTERN == 0) idx = (threadIdx.x + k * 67 + it * 131) & 8191;
            else {
                a = a * 1664525u + 1013904223u;
                idx = (a >> 12) & 8191;
            }
            if (MODE == 0) __builtin_amdgcn_ds_atomic_fadd_f64((lds_d)(buf + idx), 1.0);
            else buf[idx] = buf[idx] + 1.0;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = buf[threadIdx.x];
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <typename F>
static double mean_ticks(F launch, int blocks) {
    unsigned long long* clk;
    (void)hipMalloc(&clk, sizeof(unsigned long long) * blocks);
    launch(clk);
    launch(clk);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), clk, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += (double)v;
    (void)hipFree(clk);
    return m / blocks;
}

int main() {
    const int iters = 5000, blocks = 1024;
    double *out, *in;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    (void)hipMalloc(&in, sizeof(double) * 1024);
    std::vector<double> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0.001 * i;
    (void)hipMemcpy(in, h.data(), sizeof(double) * 1024, hipMemcpyHostToDevice);
#define RUN_FMA(MODE, NACC, WHAT)                                                                                              \
    {                                                                                                                          \
        const double t = mean_ticks([&](unsigned long long* c) { hipLaunchKernelGGL((fma3_kernel<MODE, NACC>), dim3(blocks), dim3(64), 0, 0, out, in, iters, c); }, blocks); \
        std::printf("%-44s %2d accumulators: %.3f clk per v_fma_f64 per wave\n", WHAT, NACC, t / ((double)iters * 4 * NACC));      \
    }
    RUN_FMA(1, 16, "fma(acc, s, t): one VGPR source")
    RUN_FMA(2, 16, "fma(x, s, acc): two VGPR sources")
    RUN_FMA(3, 16, "fma(x, y, acc): three VGPR sources")
    RUN_FMA(3, 8, "fma(x, y, acc): three VGPR sources")
    RUN_FMA(3, 24, "fma(x, y, acc): three VGPR sources")
#define RUN_LDS(MODE, PATTERN, THREADS, WHAT)                                                                                  \
    {                                                                                                                          \
        const double t = mean_ticks([&](unsigned long long* c) { hipLaunchKernelGGL((lds_kernel<MODE, PATTERN>), dim3(256), dim3(THREADS), 0, 0, out, 2000, c); }, 256); \
        std::printf("%-60s %3d threads/CU: %.2f clk per 64-lane update instruction per CU (%.2f lanes/clk)\n", WHAT, THREADS,    \
                    t / (2000.0 * 16 * (THREADS / 64)), 64.0 * 2000.0 * 16 * (THREADS / 64) / t);                                \
    }
    RUN_LDS(0, 0, 256, "ds_add_f64, lane-consecutive")
    RUN_LDS(0, 1, 256, "ds_add_f64, random")
    RUN_LDS(1, 0, 256, "ds_read + add + ds_write, lane-consecutive")
    RUN_LDS(1, 1, 256, "ds_read + add + ds_write, random (not atomic)")
    RUN_LDS(0, 1, 512, "ds_add_f64, random")
    RUN_LDS(0, 1, 1024, "ds_add_f64, random")
    return 0;
}
