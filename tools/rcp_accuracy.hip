// tools/rcp_accuracy.hip -- what v_rcp_f64 delivers on gfx950, and after each Newton step x <- x (2 - d x): the pivot
// reciprocals of the in-register L D L' solves (rolling_seg_dev.hpp, grouped_fused.hip) sit on the latency chain of every row.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rcp_accuracy tools/rcp_accuracy.hip && /tmp/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void rcp_kernel(const double* __restrict__ d, int n, double* __restrict__ x0, double* __restrict__ x1, double* __restrict__ x2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = d[i];
    double x = __builtin_amdgcn_rcp(v);
    x0[i] = x;
    x = x * fma(-v, x, 2.0);
    x1[i] = x;
    x = x * fma(-v, x, 2.0);
    x2[i] = x;
}

int main() {
    const int n = 1 << 22;
    std::vector<double> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;       // [0, 1)
        const int e = (int)((s >> 3) % 121) - 60;                      // 2^-60 .. 2^60
        h[i] = std::ldexp(1.0 + u, e) * ((s & 1) ? 1.0 : -1.0);
    }
    double *d, *x0, *x1, *x2;
    hipMalloc(&d, n * 8); hipMalloc(&x0, n * 8); hipMalloc(&x1, n * 8); hipMalloc(&x2, n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(rcp_kernel, dim3(n / 256), dim3(256), 0, 0, d, n, x0, x1, x2);
    std::vector<double> r0(n), r1(n), r2(n);
    hipMemcpy(r0.data(), x0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r1.data(), x1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), x2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)h[i];
        e0 = std::fmax(e0, (double)fabsl(((long double)r0[i] - t) / t));
        e1 = std::fmax(e1, (double)fabsl(((long double)r1[i] - t) / t));
        e2 = std::fmax(e2, (double)fabsl(((long double)r2[i] - t) / t));
    }
    std::printf("v_rcp_f64 max rel err %.3e (%.1f bits); +1 Newton %.3e; +2 Newton %.3e   [eps = 1.11e-16]\n", e0, -std::log2(e0), e1, e2);
    return 0;
}
