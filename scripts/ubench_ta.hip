// Cost of vector memory loads that hit in the L1 of a CU, by width, alignment and lane pattern: cycles of one CU
// per wave instruction, with 16 waves per CU issuing independent loads.  Development aid.
//   hipcc --offload-arch=gfx950 -O3 -o variants/ubench_ta scripts/ubench_ta.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v3u __attribute__((ext_vector_type(3)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) u2_a4
{
    uint32_t x, y;
};
struct __attribute__((packed, aligned(4))) u3_a4
{
    uint32_t x, y, z;
};
struct __attribute__((packed, aligned(4))) u4_a4
{
    uint32_t x, y, z, w;
};

constexpr int kLoop = 512, kUnroll = 8, kWindow = 8192; // bytes of the buffer one wave walks over

// MODE: 0 dword lane stride 4 | 1 dword stride 8 | 2 dwordx2 aligned stride 8 | 3 dwordx2 at +4 stride 8
//       4 dwordx4 aligned stride 16 | 5 dwordx4 at +4 stride 16 | 6 dwordx3 stride 12 | 7 ubyte stride 3
//       8 dwordx2 scattered (4-byte aligned) | 9 dwordx4 scattered (16-byte aligned) | 10 dword scattered
//       11 dword stride 12 | 12 ubyte stride 1 | 13 dwordx4 stride 32 | 14 dwordx2 at +4 stride 16
template <int MODE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ buf, uint32_t* out, int salt)
{
    const int lane = threadIdx.x & 63;
    unsigned off;
    switch (MODE)
    {
    case 0: off = lane * 4; break;
    case 1: off = lane * 8; break;
    case 2: off = lane * 8; break;
    case 3: off = lane * 8 + 4; break;
    case 4: off = lane * 16; break;
    case 5: off = lane * 16 + 4; break;
    case 6: off = lane * 12; break;
    case 7: off = lane * 3; break;
    case 8: off = ((lane * 2654435761u) >> 20) & 0xFFC; break;
    case 9: off = ((lane * 2654435761u) >> 20) & 0xFF0; break;
    case 10: off = ((lane * 2654435761u) >> 20) & 0xFFC; break;
    case 11: off = lane * 12; break;
    case 12: off = lane; break;
    case 13: off = lane * 32; break;
    default: off = lane * 16 + 4; break;
    }
    const char* p = buf + (threadIdx.x >> 6) * 64; // the waves of a block share the window
    uint32_t acc = 0;
    for (int it = 0; it < kLoop; ++it)
    {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
        {
            const unsigned at = off + (unsigned)(((it * kUnroll + u) * 128 + salt) & (kWindow / 2 - 1) & ~15);
            if (MODE == 0 || MODE == 1 || MODE == 10 || MODE == 11)
                acc += *(const uint32_t*)(p + at);
            else if (MODE == 2)
            {
                const v2u v = *(const v2u*)(p + at);
                acc += v.x ^ v.y;
            }
            else if (MODE == 3 || MODE == 8 || MODE == 14)
            {
                const u2_a4 v = *(const u2_a4*)(p + at);
                acc += v.x ^ v.y;
            }
            else if (MODE == 4 || MODE == 9 || MODE == 13)
            {
                const v4u v = *(const v4u*)(p + at);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
            else if (MODE == 5)
            {
                const u4_a4 v = *(const u4_a4*)(p + at);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
            else if (MODE == 6)
            {
                const u3_a4 v = *(const u3_a4*)(p + at);
                acc += v.x ^ v.y ^ v.z;
            }
            else
                acc += *(const uint8_t*)(p + at);
        }
    }
    if (acc == 0x12345u)
        out[threadIdx.x] = acc;
}

template <int MODE>
static void run(const char* what, const char* buf, uint32_t* out)
{
    const int blocks = 256 * 4; // 16 waves per CU
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, out, 0);
    hipEventRecord(a, 0);
    for (int r = 0; r < 4; ++r)
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, out, r * 16);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = 4.0 * blocks * 4 * kLoop * kUnroll / 256.0;
    const double ns = ms * 1e6 / instr_per_cu;
    std::printf("%-44s %6.2f ns per wave load per CU = %5.1f cycles at 2.1 GHz\n", what, ns, ns * 2.1);
}

int main()
{
    char* buf;
    uint32_t* out;
    hipMalloc(&buf, 1 << 20);
    hipMemset(buf, 1, 1 << 20);
    hipMalloc(&out, 4096);
    run<0>("dword, lane stride 4", buf, out);
    run<1>("dword, lane stride 8", buf, out);
    run<11>("dword, lane stride 12", buf, out);
    run<10>("dword, scattered in 4 KB", buf, out);
    run<2>("dwordx2 aligned, stride 8", buf, out);
    run<3>("dwordx2 at +4, stride 8", buf, out);
    run<14>("dwordx2 at +4, stride 16", buf, out);
    run<8>("dwordx2 scattered (4-byte aligned)", buf, out);
    run<6>("dwordx3, stride 12", buf, out);
    run<4>("dwordx4 aligned, stride 16", buf, out);
    run<5>("dwordx4 at +4, stride 16", buf, out);
    run<13>("dwordx4 aligned, stride 32", buf, out);
    run<9>("dwordx4 scattered (16-byte aligned)", buf, out);
    run<7>("ubyte, stride 3", buf, out);
    run<12>("ubyte, stride 1", buf, out);
    return 0;
}
