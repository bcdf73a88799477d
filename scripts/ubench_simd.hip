// Which SIMD of its CU each wave of a 9-wave block lands on (HW_ID register, gfx9 layout: SIMD_ID bits 5:4, CU_ID 11:8).
// Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <string>
#include <vector>
__global__ __launch_bounds__(576) void k(unsigned* out)
{
    extern __shared__ char lds[];
    lds[threadIdx.x] = 1;
    __syncthreads();
    const unsigned id = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4); // HW_ID bits 15:0
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main(int argc, char** argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 9;
    unsigned* d;
    hipMalloc(&d, 256 * 16 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 138 * 1024, 0, d);
    std::vector<unsigned> h(256 * 16);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<std::string, int> hist;
    for (int b = 0; b < 256; ++b)
    {
        std::string s;
        for (int w = 0; w < waves; ++w)
            s += char('0' + ((h[b * 16 + w] >> 4) & 3));
        hist[s]++;
    }
    for (auto& e : hist)
        std::printf("%s x%d\n", e.first.c_str(), e.second);
    return 0;
}
