// Development microbenchmark: do v_fma_f64 (vector) and v_mfma_f64_16x16x4 (matrix) instructions of TWO WAVES ON ONE SIMD run beside
// each other or take turns?  A workgroup of 512 threads: waves 0 .. 3 issue matrix instructions (4 independent accumulators), waves
// 4 .. 7 (wave w + 4 shares wave w's SIMD) issue independent vector FMAs; three launches -- matrix waves only, vector waves only, both --
// with per-wave counts chosen so that the two alone take about the same time.  together ~ max: separate pipes; together ~ sum: one FP64 unit.
// hipcc --offload-arch=gfx950 -O3 tools/fp64_share_probe.hip -o /tmp/fp64_share && /tmp/fp64_share
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void probe(int n_mfma, int n_fma, int mode, double* out) {
    const int wave = threadIdx.x >> 6;
    double r = 0.0;
    if (wave < 4) {
        if (mode & 1) {
            d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
            const double x = 1.0 + threadIdx.x * 1e-9;
            for (int i = 0; i < n_mfma; i += 4) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a3, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else if (mode & 2) {
        double v[8];
        for (int k = 0; k < 8; ++k) v[k] = 1.0 + k + threadIdx.x * 1e-9;
        const double m = 1.0000001, c = 1e-9;
        for (int i = 0; i < n_fma; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_fma(v[k], m, c);
        }
        for (int k = 0; k < 8; ++k) r += v[k];
    }
    if (r == 123.456) out[0] = r;
}
int main() {
    double* d; hipMalloc(&d, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256;  // one workgroup per CU
    auto run = [&](int nm, int nf, int mode) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, nm, nf, mode, d);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, nm, nf, mode, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3;
    };
    const int nm = 1 << 16;
    for (int ratio : {8, 16, 22, 32}) {  // vector FMAs per matrix instruction
        const int nf = nm * ratio;
        const float a = run(nm, nf, 1), b = run(nm, nf, 2), c = run(nm, nf, 3);
        printf("per wave: %d v_mfma_f64_16x16x4 | %d v_fma_f64 (x%d):  matrix alone %.3f ms (%.1f clk each at 2.4 GHz)   vector alone %.3f ms (%.2f clk each)   both %.3f ms   (max %.3f, sum %.3f)\n",
               nm, nf, ratio, a, a * 2.4e6 / nm, b, b * 2.4e6 / nf, c, a > b ? a : b, a + b);
    }
    return 0;
}
