// Development microbenchmark (companion of fp64_share_probe.hip): do 32-bit VECTOR instructions (v_fma_f32, v_and / v_lshl -- what the
// three-plane bf16 split of the wide f32 Gram executes between its matrix instructions) run beside v_mfma_f32_32x32x16_bf16 of the SIMD's
// other wave, or do they take turns?   hipcc --offload-arch=gfx950 -O3 tools/bf16_share_probe.hip -o /tmp/bf16_share && /tmp/bf16_share
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void probe(int n_mfma, int n_valu, int mode, int kind, float* out) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f16v a0 = {}, a1 = {}, a2 = {}, a3 = {};
            bf8 x;
            for (int k = 0; k < 8; ++k) x[k] = (__bf16)(1.0f + 0.001f * (threadIdx.x & 7));
            for (int i = 0; i < n_mfma; i += 4) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a3, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else if (mode & 2) {
        if (kind == 0) {
            float v[8];
            for (int k = 0; k < 8; ++k) v[k] = 1.0f + k + threadIdx.x * 1e-6f;
            const float m = 1.0000001f, c = 1e-9f;
            for (int i = 0; i < n_valu; i += 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], m, c);
            }
            for (int k = 0; k < 8; ++k) r += v[k];
        } else {
            unsigned v[8];
            for (int k = 0; k < 8; ++k) v[k] = 0x3f800000u + k + threadIdx.x;
            for (int i = 0; i < n_valu; i += 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = ((v[k] & 0xffff0000u) >> 3) + (v[k] << 1);  // (and, shift, shift-add: 32-bit integer vector work)
            }
            unsigned s = 0;
            for (int k = 0; k < 8; ++k) s += v[k];
            r = (float)s;
        }
    }
    if (r == 123.456f) out[0] = r;
}
int main() {
    float* d; hipMalloc(&d, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int nm, int nv, int mode, int kind) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, nm, nv, mode, kind, d);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, nm, nv, mode, kind, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3;
    };
    const int nm = 1 << 17;
    for (int kind = 0; kind < 2; ++kind)
        for (int ratio : {4, 8}) {
            const int nv = nm * ratio;
            const float a = run(nm, nv, 1, kind), b = run(nm, nv, 2, kind), c = run(nm, nv, 3, kind);
            printf("%s: per wave %d v_mfma_f32_32x32x16_bf16 | %d vector ops (x%d): matrix alone %.3f ms (%.1f clk each at 2.4 GHz)  vector alone %.3f ms  both %.3f ms  (max %.3f, sum %.3f)\n",
                   kind == 0 ? "v_fma_f32       " : "int and/shift/add", nm, nv, ratio, a, a * 2.4e6 / nm, b, c, a > b ? a : b, a + b);
        }
    return 0;
}
