// Does the clamp bit of VOP3P work on v_pk_add_f32 / v_pk_fma_f32 (gfx950)?  Prints the results for a few inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* out)
{
    const int t = threadIdx.x;
    v2f d = {in[2 * t], in[2 * t + 1]};
    v2f h = {0.5f, 0.5f};
    v2f r, q, u;
    asm volatile("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(d), "v"(h));
    asm volatile("v_pk_add_f32 %0, %1, 0.5 op_sel_hi:[1,0] clamp" : "=v"(q) : "v"(d));
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(u) : "v"(d), "v"(h));
    out[6 * t] = r.x, out[6 * t + 1] = r.y, out[6 * t + 2] = q.x, out[6 * t + 3] = q.y, out[6 * t + 4] = u.x, out[6 * t + 5] = u.y;
}
int main()
{
    float h_in[16] = {0.f, -0.f, 2.f, -2.f, 1e30f, -1e30f, 0.25f, -0.25f, 3.f, -7.f, 1e38f, -1e38f, 0.5f, -0.5f, __builtin_nanf(""), __builtin_inff()};
    float *in, *out, h_out[48];
    hipMalloc(&in, sizeof(h_in));
    hipMalloc(&out, sizeof(h_out));
    hipMemcpy(in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, in, out);
    hipMemcpy(h_out, out, sizeof(h_out), hipMemcpyDeviceToHost);
    for (int t = 0; t < 8; ++t)
        printf("d = (%g, %g): clamp(d + .5 vgpr) = (%g, %g); with inline 0.5 = (%g, %g); clamp(d - .5) = (%g, %g)\n", h_in[2 * t], h_in[2 * t + 1],
               h_out[6 * t], h_out[6 * t + 1], h_out[6 * t + 2], h_out[6 * t + 3], h_out[6 * t + 4], h_out[6 * t + 5]);
    return 0;
}
