// v_pk_mov_b32 operand selection on gfx950 (development aid): which halves of the two sources land where.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(float* out)
{
    v2f a = v2f{1.0f + threadIdx.x, 2.0f}, b = v2f{3.0f, 4.0f};
    asm volatile("" : "+v"(a), "+v"(b));
    v2f r0, r1, r2, r3;
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(r0) : "v"(a), "v"(b));
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r1) : "v"(a), "v"(b));
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(r2) : "v"(a), "v"(b));
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(r3) : "v"(a), "v"(b));
    if (threadIdx.x == 0)
    {
        out[0] = r0.x, out[1] = r0.y, out[2] = r1.x, out[3] = r1.y;
        out[4] = r2.x, out[5] = r2.y, out[6] = r3.x, out[7] = r3.y;
    }
}
int main()
{
    float* d;
    if (hipMalloc(&d, 32) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[8];
    if (hipMemcpy(h, d, 32, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const char* n[4] = {"[0,0]", "[1,0]", "[0,1]", "[1,1]"};
    for (int i = 0; i < 4; ++i)
        std::printf("a=(1,2) b=(3,4) op_sel:%s -> (%g, %g)\n", n[i], h[2 * i], h[2 * i + 1]);
    return 0;
}
