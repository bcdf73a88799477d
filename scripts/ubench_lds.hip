// LDS read cost by access shape (development aid): cycles per wave-level read instruction, one block of 9 waves per CU
// reading a 64 KB array.  Cases: ds_read_b128 at 16-byte aligned addresses and at addresses 4, 8, 12 bytes off
// (neighbouring lanes 16 bytes apart), ds_read_b32 with neighbouring lanes 4 and 16 bytes apart.
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench_lds.hip -o build/ubench_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const char lds_cchar;

template <int BYTES> __device__ __forceinline__ float read_at(unsigned at);
template <> __device__ __forceinline__ float read_at<16>(unsigned at)
{
    v4f v;
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(at) : "memory");
    return v.x + v.w;
}
template <> __device__ __forceinline__ float read_at<4>(unsigned at)
{
    float v;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(at) : "memory");
    return v;
}

// `reads` dependent-free reads per lane; lane address = lane * stride + off + a walk over the array
template <int BYTES> __global__ __launch_bounds__(576) void k(float* out, unsigned long long* ticks, int stride, int off, int reads)
{
    __shared__ __attribute__((aligned(16))) float arr[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x)
        arr[i] = (float)i;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned at = lane * stride + off + wave * 4096;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reads; i += 4)
    {
        // four reads in flight
        float a0, a1, a2, a3;
        if (BYTES == 16)
        {
            v4f v0, v1, v2, v3;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(at) : "memory");
            a0 = v0.x, a1 = v1.y, a2 = v2.z, a3 = v3.w;
        }
        else
        {
            asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:1024\n ds_read_b32 %2, %4 offset:2048\n ds_read_b32 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(at) : "memory");
        }
        acc += a0 + a1 + a2 + a3;
        at = (at + 4096) & 0xFFFF;
        at = at > 60000 ? at - 8192 : at;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0)
        ticks[blockIdx.x] = t1 - t0;
}

template <int BYTES> static void run(const char* name, int stride, int off, float* d, unsigned long long* t)
{
    const int reads = 4096;
    hipLaunchKernelGGL(k<BYTES>, dim3(256), dim3(576), 0, 0, d, t, stride, off, reads);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<BYTES>, dim3(256), dim3(576), 0, 0, d, t, stride, off, reads);
    if (hipDeviceSynchronize() != hipSuccess)
    {
        std::printf("%s: failed\n", name);
        return;
    }
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), t, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h)
        mean += (double)v;
    mean /= 256;
    // 9 waves share the CU's LDS: cycles of the LDS pipe per wave-level instruction
    std::printf("%-46s %.1f cycles per read per wave (9 waves per CU)\n", name, mean / reads / 9.0);
}

int main()
{
    float* d;
    unsigned long long* t;
    hipMalloc(&d, 256 * 576 * 4);
    hipMalloc(&t, 256 * 8);
    run<16>("b128, lanes 16 B apart, aligned", 16, 0, d, t);
    run<16>("b128, lanes 16 B apart, 4 bytes off", 16, 4, d, t);
    run<16>("b128, lanes 16 B apart, 8 bytes off", 16, 8, d, t);
    run<16>("b128, lanes 16 B apart, 12 bytes off", 16, 12, d, t);
    run<16>("b128, lanes 20 B apart (unaligned, drifting)", 20, 0, d, t);
    run<4>("b32, lanes 4 B apart", 4, 0, d, t);
    run<4>("b32, lanes 16 B apart", 16, 0, d, t);
    run<4>("b32, lanes 16 B apart, 4 bytes off", 16, 4, d, t);
    return 0;
}
