// tools/valu_f64_rate.hip -- what one wave can do with v_fma_f64 on gfx950: issue rate with N independent accumulators, latency
// of a dependent chain, the same with 1 / 2 / 4 waves per SIMD, and the cost of v_accvgpr moves and v_rcp_f64 in the mix.  The
// in-register solves of rolling_seg_dev.hpp / grouped_fused.hip run at ONE or TWO waves per SIMD: their time is set by these numbers.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu tools/valu_f64_rate.hip && /tmp/valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NACC>
__global__ void fma_kernel(double* out, int iters, double a, double b, unsigned long long* clk) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (double)threadIdx.x + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

__global__ void rcp_chain_kernel(double* out, int iters, double a, unsigned long long* clk) {
    double d = 1.5 + threadIdx.x * 1e-3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        double x = __builtin_amdgcn_rcp(d);
        x = x * __builtin_fma(-d, x, 2.0);
        x = x * __builtin_fma(-d, x, 2.0);
        d = __builtin_fma(x, a, 1.25);  // next "pivot" depends on the reciprocal
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = d;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NACC>
static void run(const char* what, int waves_per_simd, int iters) {
    // one workgroup of 64 * 4 * waves_per_simd threads per CU would spread over the 4 SIMDs; use single-wave workgroups and enough of
    // them to put `waves_per_simd` on every SIMD of every CU (256 CUs x 4 SIMDs)
    const int blocks = 256 * 4 * waves_per_simd;
    double* out;
    unsigned long long* clk;
    (void)hipMalloc(&out, sizeof(double) * blocks * 64);
    (void)hipMalloc(&clk, sizeof(unsigned long long) * blocks);
    hipLaunchKernelGGL((fma_kernel<NACC>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9, clk);
    hipLaunchKernelGGL((fma_kernel<NACC>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9, clk);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), clk, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= blocks;
    // s_memtime ticks at 100 MHz: convert with the shader clock measured by wall time is overkill here -- report ticks per FMA
    std::printf("%-34s %d wave(s)/SIMD, %2d independent accumulators: %.4f memtime ticks per v_fma_f64 per wave\n", what, waves_per_simd, NACC,
                mean / ((double)iters * NACC));
    (void)hipFree(out);
    (void)hipFree(clk);
}

int main() {
    const int iters = 20000;
    run<1>("dependent chain", 1, iters);
    run<2>("2 chains", 1, iters);
    run<4>("4 chains", 1, iters);
    run<8>("8 chains", 1, iters);
    run<16>("16 chains", 1, iters);
    run<32>("32 chains", 1, iters);
    run<1>("dependent chain", 2, iters);
    run<4>("4 chains", 2, iters);
    run<16>("16 chains", 2, iters);
    run<16>("16 chains", 4, iters);
    {
        const int blocks = 1024;
        double* out;
        unsigned long long* clk;
        (void)hipMalloc(&out, sizeof(double) * blocks * 64);
        (void)hipMalloc(&clk, sizeof(unsigned long long) * blocks);
        hipLaunchKernelGGL(rcp_chain_kernel, dim3(blocks), dim3(64), 0, 0, out, iters, 0.3, clk);
        hipLaunchKernelGGL(rcp_chain_kernel, dim3(blocks), dim3(64), 0, 0, out, iters, 0.3, clk);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), clk, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        std::printf("pivot chain (rcp + 2 Newton + 1 fma = 6 dependent ops), 1 wave/SIMD: %.4f memtime ticks per link\n", mean / blocks / iters);
        // the tick: time a known-length kernel
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rcp_chain_kernel, dim3(blocks), dim3(64), 0, 0, out, iters, 0.3, clk);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h.data(), clk, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        std::printf("kernel %.3f ms for %llu memtime ticks -> one tick = %.2f ns\n", ms, h[0], ms * 1e6 / (double)h[0]);
    }
    return 0;
}
