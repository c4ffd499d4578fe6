// Where do the waves of a 512-thread workgroup with the whole LDS of a CU land?  Prints (workgroup, wave in workgroup) -> XCC / SE / CU / SIMD
// from HW_REG_HW_ID / HW_REG_XCC_ID.  The paired grouped stream (grouped_mid.hip, PAIRED) relies on waves w and w + 4 sharing a SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/wave_placement.hip -o /tmp/wave_placement && /tmp/wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512) void where_kernel(unsigned* out) {
    extern __shared__ char lds[];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID, all 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
        out[(blockIdx.x * 8 + wv) * 2] = hw;
        out[(blockIdx.x * 8 + wv) * 2 + 1] = xcc;
    }
    lds[threadIdx.x] = 0;
    __syncthreads();
    // stay resident long enough for every workgroup of the grid to be placed
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000000) __builtin_amdgcn_s_sleep(10);
}

int main() {
    const int wgs = 256, lds = 4 * 40608;
    unsigned* d = nullptr;
    hipMalloc(&d, wgs * 8 * 2 * sizeof(unsigned));
    hipFuncSetAttribute(reinterpret_cast<const void*>(where_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(where_kernel, dim3(wgs), dim3(512), lds, 0, d);
    std::vector<unsigned> h(wgs * 8 * 2);
    if (hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) { std::printf("launch failed\n"); return 1; }
    int same = 0, diff = 0, hist[4][2] = {};
    for (int b = 0; b < wgs; ++b) {
        for (int w = 0; w < 4; ++w) {
            const unsigned a = h[(b * 8 + w) * 2], c = h[(b * 8 + w + 4) * 2];
            const unsigned sa = (a >> 4) & 3, sc = (c >> 4) & 3, cua = (a >> 8) & 15, cuc = (c >> 8) & 15;
            if (sa == sc && cua == cuc) ++same; else ++diff;
            ++hist[sa][0];
            ++hist[sc][1];
        }
        if (b < 4) {
            std::printf("workgroup %d:", b);
            for (int w = 0; w < 8; ++w) {
                const unsigned a = h[(b * 8 + w) * 2];
                std::printf("  w%d: xcc %u se %u cu %u simd %u slot %u", w, h[(b * 8 + w) * 2 + 1] & 15, (a >> 13) & 7, (a >> 8) & 15, (a >> 4) & 3, a & 15);
            }
            std::printf("\n");
        }
    }
    std::printf("pairs (w, w + 4) on the same SIMD: %d, on different SIMDs: %d\n", same, diff);
    for (int s = 0; s < 4; ++s) std::printf("SIMD %d: %d streaming waves (0 .. 3), %d solving waves (4 .. 7)\n", s, hist[s][0], hist[s][1]);
    return 0;
}
