// Development microbenchmark: achievable streaming-READ bandwidth on this box, to price the Gram kernel against.
// hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o tools/membw && tools/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int UNROLL>
__global__ __launch_bounds__(256) void read_sum(const d2* __restrict__ p, size_t nvec, double* out) {
    double acc = 0;
    size_t i = (size_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x * UNROLL;
    for (; i + (UNROLL - 1) * blockDim.x < nvec; i += stride) {
        d2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u][0] + v[u][1];
    }
    if (acc == 123.456) out[0] = acc;
}
// 17 column streams, one 1 KiB piece of each per wave iteration (the Gram kernel's access pattern, no math)
__global__ __launch_bounds__(256) void read_cols(const double* const* __restrict__ cols, int nc, size_t n, double* out) {
    const int lane = threadIdx.x & 63;
    const size_t wid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
    const size_t ntiles = n / 128;
    double acc = 0;
    for (size_t t = wid; t < ntiles; t += nw) {
        d2 v[17];
#pragma unroll
        for (int c = 0; c < 17; ++c) if (c < nc) v[c] = *reinterpret_cast<const d2*>(cols[c] + t * 128 + lane * 2);
#pragma unroll
        for (int c = 0; c < 17; ++c) if (c < nc) acc += v[c][0] + v[c][1];
    }
    if (acc == 123.456) out[0] = acc;
}
// the grouped kernel's ownership: wave w streams the contiguous row range [w N / W, (w+1) N / W) of every column, K
// consecutive 1 KiB pieces per column and iteration (K = 1 is the kernel's tile); NT: non-temporal loads
template <int K, bool NT>
__global__ __launch_bounds__(64) void read_cols_owned(const double* const* __restrict__ cols, int nc, size_t n, double* out) {
    const int lane = threadIdx.x & 63;
    const size_t W = gridDim.x, w = blockIdx.x;
    const size_t ntiles = n / (128 * K);
    const size_t t0 = ntiles * w / W, t1 = ntiles * (w + 1) / W;
    double acc = 0;
    for (size_t t = t0; t < t1; ++t) {
        d2 v[17][K];
#pragma unroll
        for (int c = 0; c < 17; ++c)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const d2* q = reinterpret_cast<const d2*>(cols[c] + (t * K + k) * 128 + lane * 2);
                if (c < nc) v[c][k] = NT ? __builtin_nontemporal_load(q) : *q;
            }
#pragma unroll
        for (int c = 0; c < 17; ++c)
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (c < nc) acc += v[c][k][0] + v[c][k][1];
    }
    if (acc == 123.456) out[0] = acc;
}
int main() {
    const size_t n = 100000000, nc = 17;
    std::vector<double*> cols(nc);
    for (auto& c : cols) { hipMalloc(&c, n * 8); hipMemset(c, 1, n * 8); }
    double** dcols; hipMalloc(&dcols, nc * 8); hipMemcpy(dcols, cols.data(), nc * 8, hipMemcpyHostToDevice);
    double* out; hipMalloc(&out, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](auto f, const char* name, double bytes) {
        f(); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-40s %.3f ms  %.1f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    for (int blocks : {512, 1024, 2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "read_sum<4> one column, %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(read_sum<4>, dim3(blocks), dim3(256), 0, 0, (const d2*)cols[0], n / 2, out); }, nm, n * 8.0);
        snprintf(nm, 64, "read_sum<8> one column, %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(read_sum<8>, dim3(blocks), dim3(256), 0, 0, (const d2*)cols[0], n / 2, out); }, nm, n * 8.0);
    }
    for (int blocks : {512, 1024, 2048, 4096}) {
        char nm[64];
        snprintf(nm, 64, "read_cols 17 streams, %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(read_cols, dim3(blocks), dim3(256), 0, 0, dcols, (int)nc, n, out); }, nm, n * 8.0 * nc);
    }
    for (int blocks : {2048, 4096}) {
        char nm[96];
        auto run = [&](auto k_c, auto nt_c, const char* tag) {
            snprintf(nm, 96, "owned ranges K=%d %s, %d waves", decltype(k_c)::value, tag, blocks);
            time([&] { hipLaunchKernelGGL((read_cols_owned<decltype(k_c)::value, decltype(nt_c)::value>), dim3(blocks), dim3(64), 0, 0, dcols, (int)nc, n, out); }, nm, n * 8.0 * nc);
        };
        run(std::integral_constant<int, 1>{}, std::false_type{}, "");
        run(std::integral_constant<int, 2>{}, std::false_type{}, "");
        run(std::integral_constant<int, 4>{}, std::false_type{}, "");
        run(std::integral_constant<int, 1>{}, std::true_type{}, "nt");
        run(std::integral_constant<int, 2>{}, std::true_type{}, "nt");
    }
    return 0;
}
