// Device-to-device copy variants, 1 GiB, read + write GB/s: which shape of a plain f4 copy gets closest to
// what the memory system delivers.  Development aid; build: hipcc --offload-arch=gfx950 -O3 -o variants/ubench_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_stride(const f4* __restrict__ s, f4* __restrict__ d, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride)
        d[e] = s[e];
}
__global__ __launch_bounds__(256) void k_one(const f4* __restrict__ s, f4* __restrict__ d, size_t n)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n)
        d[e] = s[e];
}
template <int U, bool NT> __global__ __launch_bounds__(256) void k_unroll(const f4* __restrict__ s, f4* __restrict__ d, size_t n)
{
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * 256 < n)
            v[u] = NT ? __builtin_nontemporal_load(s + base + u * 256) : s[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * 256 < n)
        {
            if (NT)
                __builtin_nontemporal_store(v[u], d + base + u * 256);
            else
                d[base + u * 256] = v[u];
        }
}
template <int U, bool NT> __global__ __launch_bounds__(256) void k_stride_unroll(const f4* __restrict__ s, f4* __restrict__ d, size_t n)
{
    const size_t step = (size_t)gridDim.x * 256 * U;
    for (size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x; base < n; base += step)
    {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + u * 256 < n)
                v[u] = NT ? __builtin_nontemporal_load(s + base + u * 256) : s[base + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + u * 256 < n)
            {
                if (NT)
                    __builtin_nontemporal_store(v[u], d + base + u * 256);
                else
                    d[base + u * 256] = v[u];
            }
    }
}

template <typename F> static void run(const char* name, F launch, size_t bytes)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i)
        launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int i = 0; i < reps; ++i)
        launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::printf("%-34s %8.1f GB/s\n", name, 2.0 * bytes * reps / (ms * 1e-3) / 1e9);
}

int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    f4 *s, *d;
    CK(hipMalloc(&s, bytes));
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(s, 1, bytes));
    CK(hipMemset(d, 0, bytes));
    run("grid-stride 8192 blocks", [&] { hipLaunchKernelGGL(k_stride, dim3(8192), dim3(256), 0, 0, s, d, n); }, bytes);
    run("grid-stride 2048 blocks", [&] { hipLaunchKernelGGL(k_stride, dim3(2048), dim3(256), 0, 0, s, d, n); }, bytes);
    run("one f4 per thread", [&] { hipLaunchKernelGGL(k_one, dim3((unsigned)(n / 256)), dim3(256), 0, 0, s, d, n); }, bytes);
    run("4 per thread", [&] { hipLaunchKernelGGL((k_unroll<4, false>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, s, d, n); }, bytes);
    run("4 per thread, nontemporal", [&] { hipLaunchKernelGGL((k_unroll<4, true>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, s, d, n); }, bytes);
    run("8 per thread", [&] { hipLaunchKernelGGL((k_unroll<8, false>), dim3((unsigned)(n / 2048)), dim3(256), 0, 0, s, d, n); }, bytes);
    run("8 per thread, nontemporal", [&] { hipLaunchKernelGGL((k_unroll<8, true>), dim3((unsigned)(n / 2048)), dim3(256), 0, 0, s, d, n); }, bytes);
    run("one per thread, nontemporal", [&] { hipLaunchKernelGGL((k_unroll<1, true>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, s, d, n); }, bytes);
    run("stride 4x, 2048 blocks", [&] { hipLaunchKernelGGL((k_stride_unroll<4, false>), dim3(2048), dim3(256), 0, 0, s, d, n); }, bytes);
    run("stride 4x nt, 2048 blocks", [&] { hipLaunchKernelGGL((k_stride_unroll<4, true>), dim3(2048), dim3(256), 0, 0, s, d, n); }, bytes);
    run("stride 4x nt, 4096 blocks", [&] { hipLaunchKernelGGL((k_stride_unroll<4, true>), dim3(4096), dim3(256), 0, 0, s, d, n); }, bytes);
    run("stride 2x nt, 8192 blocks", [&] { hipLaunchKernelGGL((k_stride_unroll<2, true>), dim3(8192), dim3(256), 0, 0, s, d, n); }, bytes);
    run("hipMemcpyDtoD", [&] { CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0)); }, bytes);
    return 0;
}
