// Development microbenchmark: can the matrix core's operand layout be loaded STRAIGHT from column buffers (no LDS tile)?
// Lane (f = lane & 15, q = lane >> 4) loads 16 bytes of column f: rows 8 s + 2 q, 8 s + 2 q + 1 -- per instruction 16 columns x
// 64 contiguous bytes (the next instruction takes the other half of the same 128-byte lines).  y: 16 bytes per lane, 16-fold
// redundant.  No math (a checksum): what the access pattern alone delivers, by waves per CU and steps in flight.
// hipcc --offload-arch=gfx950 -O3 tools/membw_operand.hip -o /tmp/membw_operand && /tmp/membw_operand
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT, bool WITHY>
__global__ __launch_bounds__(64) void read_operand(const double* const* __restrict__ cols, size_t n, double* out) {
    const int lane = threadIdx.x & 63, f = lane & 15, q = lane >> 4;
    const size_t W = gridDim.x, w = blockIdx.x;
    const size_t nsteps = n / 8;  // 8 rows per load step
    const size_t s0 = nsteps * w / W, s1 = nsteps * (w + 1) / W;
    const double* col = cols[f];
    const double* ycol = cols[16];
    double acc = 0;
    for (size_t s = s0; s + U <= s1; s += U) {
        d2 v[U], yv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const d2* p = reinterpret_cast<const d2*>(col + (s + u) * 8 + 2 * q);
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
            if (WITHY) {
                const d2* py = reinterpret_cast<const d2*>(ycol + (s + u) * 8 + 2 * q);
                yv[u] = NT ? __builtin_nontemporal_load(py) : *py;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc += v[u][0] + v[u][1];
            if (WITHY) acc += yv[u][0] * yv[u][1];
        }
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
        printf("%-64s %.3f ms  %.1f GB/s  %.3f of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000);
    };
    for (int per_cu : {8, 16, 24, 32}) {
        const int waves = 256 * per_cu;
        char nm[128];
        auto run = [&](auto u_c, auto nt_c, auto y_c) {
            constexpr int U = decltype(u_c)::value;
            constexpr bool NT = decltype(nt_c)::value, Y = decltype(y_c)::value;
            snprintf(nm, 128, "operand layout, %d waves/CU, %d steps in flight%s%s", per_cu, U, NT ? ", nt" : "", Y ? ", +y" : "");
            time([&] { hipLaunchKernelGGL((read_operand<U, NT, Y>), dim3(waves), dim3(64), 0, 0, dcols, n, out); }, nm,
                 n * 8.0 * (Y ? 17 : 16));
        };
        run(std::integral_constant<int, 4>{}, std::true_type{}, std::false_type{});
        run(std::integral_constant<int, 8>{}, std::true_type{}, std::false_type{});
        run(std::integral_constant<int, 16>{}, std::true_type{}, std::false_type{});
        run(std::integral_constant<int, 8>{}, std::false_type{}, std::false_type{});
        run(std::integral_constant<int, 8>{}, std::true_type{}, std::true_type{});
        run(std::integral_constant<int, 16>{}, std::true_type{}, std::true_type{});
    }
    return 0;
}
