// Layout and rate of v_mfma_f64_4x4x4_4b_f64 on gfx950 (four independent 4x4x4 products per instruction, one f64 per lane for A, B and D).
// For every pair of lanes (la, lb): A = 1 in lane la, B = 1 in lane lb -> which lane of D becomes 1?  From the table: A lane = 16 b + 4 k + i,
// B lane = 16 b + 4 k + j, D lane = 16 b + 4 i + j ... printed as found.  Then the issue rate against v_mfma_f64_16x16x4_f64.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_4x4_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe(int* out) {  // out[la * 64 + lb] = lane of D that is 1 (or -1), one wave
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) out[la * 64 + lb] = m ? __ffsll((long long)m) - 1 : -1;
        }
}
__global__ void rate(double* out, long long* clk, int iters) {
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane, b = 0.5 * lane;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    long long t0 = wall_clock64();
    long long s0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    long long s1 = __builtin_amdgcn_s_memtime();
    d4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0}, e2 = {0, 0, 0, 0}, e3 = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e0, 0, 0, 0);
        e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e1, 0, 0, 0);
        e2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e2, 0, 0, 0);
        e3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e3, 0, 0, 0);
    }
    long long s2 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + e0[0] + e1[1] + e2[2] + e3[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = s1 - s0; clk[1] = s2 - s1; clk[2] = wall_clock64() - t0; }
}
int main() {
    int* d; hipMalloc(&d, 64 * 64 * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    std::vector<int> h(64 * 64);
    hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    int hits = 0, okf = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb)
            if (h[la * 64 + lb] >= 0) {
                ++hits;
                // hypothesis: A lane = 16 b + 4 k + i, B lane = 16 b + 4 k + j, D lane = 16 b + 4 i + j   (i, j, k in 0..3)
                const int ba = la / 16, ka = (la / 4) % 4, ia = la % 4, bb = lb / 16, kb = (lb / 4) % 4, jb = lb % 4;
                if (ba == bb && ka == kb && h[la * 64 + lb] == 16 * ba + 4 * ia + jb) ++okf;
            }
    std::printf("pairs with a non-zero product: %d (4 blocks x 4 k x 4 i x 4 j = 256 expected); matching  A = 16b + 4k + i, B = 16b + 4k + j, D = 16b + 4i + j: %d\n", hits, okf);
    if (okf != hits) {
        std::printf("first pairs: ");
        int n = 0;
        for (int la = 0; la < 64 && n < 24; ++la)
            for (int lb = 0; lb < 64 && n < 24; ++lb)
                if (h[la * 64 + lb] >= 0) { std::printf("(%d,%d)->%d ", la, lb, h[la * 64 + lb]); ++n; }
        std::printf("\n");
    }
    double* o; long long* c; hipMalloc(&o, 1024 * 256 * 8); hipMalloc(&c, 3 * 8);
    const int iters = 20000;
    hipLaunchKernelGGL(rate, dim3(1024), dim3(256), 0, 0, o, c, iters);
    long long hc[3]; hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
    std::printf("one wave per SIMD, four independent accumulators: %.1f memtime ticks per v_mfma_f64_4x4x4_4b, %.1f per v_mfma_f64_16x16x4\n", (double)hc[0] / (4.0 * iters),
                (double)hc[1] / (4.0 * iters));
    return 0;
}
