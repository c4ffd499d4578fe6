// Development microbenchmark: what does ONE wave per SIMD pay per 4-row step of the 17 .. 32-feature grouped stream when nothing but the
// matrix instructions is there?  Patterns (operands in registers, no memory):
//   0: three v_mfma_f64_16x16x4 per step on three accumulators (acc0 += a a, acc1 += a b, acc2 += b b): the 19 .. 32-feature step
//   1: one v_mfma_f64_16x16x4 + two v_mfma_f64_4x4x4_4b (the 17 / 18-feature step)
//   2: pattern 0 with SIX accumulators (even / odd steps on their own sets)
//   3: pattern 0, a fresh operand pair per step from a register ring of 16 (as the direct form's unrolled run reads them)
//   4: one v_mfma_f64_16x16x4 per step on ONE accumulator (the <= 16-feature step): the dependent-issue interval
//   5: one v_mfma_f64_16x16x4 per step alternating TWO accumulators
//   6 / 7: patterns 0 / 1 with the direct form's memory stream beside them: sixteen 1 KiB loads per sixteen steps into a register set, waited for
//          (and folded into one register) in front of the next sixteen
// s_memtime around the loop of every wave (max over waves) and the wall time.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_chain_probe.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define M16(a, b, c) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
#define M4(a, b, c) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int PAT, int NL = 16>
__global__ __launch_bounds__(512) void probe_mem(int steps, double* out, unsigned long long* clk, const u4* __restrict__ mem) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) return;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
    double q1 = 0.0, q2 = 0.0;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const u4* src = mem + ((size_t)(blockIdx.x * 4 + wave) * (steps / 16)) * (64 * NL) + lane;  // NL KiB per sixteen steps and wave
    u4 nxt[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) nxt[k] = __builtin_nontemporal_load(src + 64 * k);
    unsigned fold = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s += 16) {
        u4 cur[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            cur[k] = nxt[k];
            asm volatile("" : "+v"(cur[k]));
        }
        src += 64 * NL;
        if (s + 16 < steps) {
#pragma unroll
            for (int k = 0; k < NL; ++k) nxt[k] = __builtin_nontemporal_load(src + 64 * k);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (PAT == 6) { M16(a, a, c0); M16(a, b, c1); M16(b, b, c2); }
            if constexpr (PAT == 7) { M16(a, a, c0); M4(a, b, q1); M4(b, b, q2); }
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) fold ^= cur[k][0];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const double r = c0[0] + c1[1] + c2[2] + q1 + q2 + (double)fold;
    if (r == 123.456) out[0] = r;
    if ((threadIdx.x & 63) == 0) atomicMax(clk, t1 - t0);
}
template <int PAT>
__global__ __launch_bounds__(512) void probe(int steps, double* out, unsigned long long* clk) {
    const int wave = threadIdx.x >> 6;
    if (wave >= 4) return;  // waves 4 .. 7 would be the solving waves: idle here
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
    double q1 = 0.0, q2 = 0.0;
    double ra[16], rb[16];
    for (int k = 0; k < 16; ++k) {
        ra[k] = 1.0 + (threadIdx.x + k) * 1e-9;
        rb[k] = 1.0 - (threadIdx.x + k) * 1e-9;
    }
    const double a = ra[0], b = rb[0];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (PAT == 0) { M16(a, a, c0); M16(a, b, c1); M16(b, b, c2); }
            if constexpr (PAT == 1) { M16(a, a, c0); M4(a, b, q1); M4(b, b, q2); }
            if constexpr (PAT == 2) {
                if (u & 1) { M16(a, a, c3); M16(a, b, c4); M16(b, b, c5); }
                else { M16(a, a, c0); M16(a, b, c1); M16(b, b, c2); }
            }
            if constexpr (PAT == 3) { M16(ra[u], ra[u], c0); M16(ra[u], rb[u], c1); M16(rb[u], rb[u], c2); }
            if constexpr (PAT == 4) { M16(a, a, c0); }
            if constexpr (PAT == 5) {
                if (u & 1) M16(a, a, c1);
                else M16(a, a, c0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const double r = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + q1 + q2;
    if (r == 123.456) out[0] = r;
    if ((threadIdx.x & 63) == 0) atomicMax(clk, t1 - t0);
}
template <int PAT>
void run(const char* name, int per_step16, int per_step4) {
    double* d; unsigned long long* c;
    hipMalloc(&d, 8); hipMalloc(&c, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int steps = 1 << 16;
    hipLaunchKernelGGL(probe<PAT>, dim3(256), dim3(512), 0, 0, steps, d, c);
    hipDeviceSynchronize();
    hipMemset(c, 0, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<PAT>, dim3(256), dim3(512), 0, 0, steps, d, c);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h = 0; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-58s %8.1f ticks / step  (%d x 16x16x4 + %d x 4x4x4_4b)   wall %.3f ms = %.1f clk / step at 2.4 GHz;  ticks / wall = %.2f GHz\n", name, (double)h / steps,
           per_step16, per_step4, ms, ms * 2.4e6 / steps, (double)h / ms / 1e6);
    hipFree(d); hipFree(c);
}
template <int PAT, int NL>
void run_mem(const char* name) {
    double* d; unsigned long long* c; u4* mem;
    const int steps = 1 << 13;
    const size_t bytes = (size_t)1024 * (steps / 16) * 1024 * NL;
    hipMalloc(&d, 8); hipMalloc(&c, 8);
    if (hipMalloc(&mem, bytes) != hipSuccess) { printf("no memory\n"); return; }
    hipMemset(mem, 0x3c, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe_mem<PAT, NL>), dim3(256), dim3(512), 0, 0, steps, d, c, mem);
    hipDeviceSynchronize();
    hipMemset(c, 0, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_mem<PAT, NL>), dim3(256), dim3(512), 0, 0, steps, d, c, mem);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h = 0; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-58s %8.1f ticks / step   wall %.3f ms = %.1f clk / step at 2.4 GHz;  %.2f TB/s;  ticks / wall = %.2f GHz\n", name, (double)h / steps, ms, ms * 2.4e6 / steps,
           bytes / ms / 1e9, (double)h / ms / 1e6);
    hipFree(d); hipFree(c); hipFree(mem);
}
int main() {
    run_mem<6, 16>("6: three 16x16x4 + 16 KiB of loads per sixteen steps");
    run_mem<6, 8>("6: three 16x16x4 +  8 KiB of loads per sixteen steps");
    run_mem<6, 4>("6: three 16x16x4 +  4 KiB of loads per sixteen steps");
    run_mem<7, 16>("7: one 16x16x4 + two 4x4x4_4b + 16 KiB per sixteen steps");
    run_mem<7, 8>("7: one 16x16x4 + two 4x4x4_4b +  8 KiB per sixteen steps");
    run_mem<7, 4>("7: one 16x16x4 + two 4x4x4_4b +  4 KiB per sixteen steps");
    run_mem<7, 2>("7: one 16x16x4 + two 4x4x4_4b +  2 KiB per sixteen steps");
    run<0>("0: three 16x16x4, three accumulators", 3, 0);
    run<1>("1: one 16x16x4 + two 4x4x4_4b", 1, 2);
    run<2>("2: three 16x16x4, six accumulators (even / odd steps)", 3, 0);
    run<3>("3: three 16x16x4, operands from a ring of 16 pairs", 3, 0);
    run<4>("4: one 16x16x4, one accumulator", 1, 0);
    run<5>("5: one 16x16x4, two accumulators alternating", 1, 0);
    return 0;
}
