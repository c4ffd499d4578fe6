// mfma_peak.hip -- what the matrix cores sustain on this part (clock under load included): back-to-back
// v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64 on 4 independent accumulators per wave, no memory traffic.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak.bin
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef double d4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_f32(float* out, int iters, float a0) {
    f16v acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x, b = a0 - threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_f64(double* out, int iters, double a0) {
    d4v acc[8];
    for (int t = 0; t < 8; ++t)
        for (int r = 0; r < 4; ++r) acc[t][r] = 0.;
    double a = a0 + threadIdx.x, b = a0 - threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
    for (int t = 0; t < 8; ++t)
        for (int r = 0; r < 4; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    void* out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wps = 1; wps <= 3; ++wps) {  // waves per SIMD
        for (int which = 0; which < 2; ++which) {
            const int iters = which == 0 ? 20000 : 40000;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k_f32, dim3(cus * wps), dim3(256), 0, 0, (float*)out, iters, 1.0f);
                else hipLaunchKernelGGL(k_f64, dim3(cus * wps), dim3(256), 0, 0, (double*)out, iters, 1.0);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double n_mfma = (double)cus * wps * 4 * iters * 16;
                const double flop = which == 0 ? 32. * 32 * 2 * 2 : 16. * 16 * 4 * 2;
                if (rep == 2)
                    printf("%s waves/SIMD=%d  %.2f ms  %.1f TFLOP/s  (%.1f clk/MFMA/SIMD at 2.4 GHz)\n",
                           which == 0 ? "f32 32x32x2" : "f64 16x16x4", wps, ms, n_mfma * flop / (ms * 1e-3) / 1e12,
                           ms * 1e-3 * 2.4e9 / (iters * 16.0 * wps));
            }
        }
    }
    return 0;
}
