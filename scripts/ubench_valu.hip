// Issue rate of vector-ALU instructions on gfx950 as a function of the waves per SIMD.
// Every wave runs REPS trips of 64 independent instructions of one kind (16 registers x 4);
// blocks of 256 threads (one wave per SIMD), `bpc` blocks per CU.  Prints SIMD cycles per
// wave-instruction from the wall time at the nominal 2.4 GHz (the real clock under load is lower).
// Build: hipcc --offload-arch=gfx950 -O3 -o variants/ubench_valu scripts/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int REPS = 1000;
typedef float v2f __attribute__((ext_vector_type(2)));

#define OP16(A) A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11) A(12) A(13) A(14) A(15)
#define OP64(A) OP16(A) OP16(A) OP16(A) OP16(A)

struct Regs
{
    float a[16], b, c;
    double d[16], db, dc;
    v2f p[16], pb, pc;
    unsigned u[16], ub, uc;
    unsigned long long q[16];
};

#define KERNEL(NAME, STMT)                                                                       \
    __global__ __launch_bounds__(256) void NAME(float* out)                                      \
    {                                                                                            \
        Regs r;                                                                                  \
        r.b = 1.0001f + threadIdx.x * 1e-6f, r.c = 0.5f, r.db = 1.0000001, r.dc = 0.25;          \
        r.pb = v2f{r.b, r.b}, r.pc = v2f{r.c, r.c}, r.ub = threadIdx.x | 1, r.uc = 3;            \
        const unsigned sk = __builtin_amdgcn_readfirstlane(0xC0000000u + blockIdx.x * 0);        \
        for (int i = 0; i < 16; ++i)                                                             \
            r.a[i] = threadIdx.x + i, r.d[i] = r.a[i], r.p[i] = v2f{r.a[i], r.a[i]},             \
            r.u[i] = threadIdx.x + i, r.q[i] = r.u[i];                                           \
        for (int t = 0; t < REPS; ++t)                                                           \
        {                                                                                        \
            OP64(STMT)                                                                           \
        }                                                                                        \
        float s = 0;                                                                             \
        for (int i = 0; i < 16; ++i)                                                             \
            s += r.a[i] + (float)r.d[i] + r.p[i].x + r.p[i].y + (float)r.u[i] + (float)r.q[i];   \
        out[blockIdx.x * 256 + threadIdx.x] = s + (float)sk;                                     \
    }

#define S_fma(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r.a[i]) : "v"(r.b), "v"(r.c));
#define S_fmac(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r.a[i]) : "v"(r.b), "v"(r.c));
#define S_fma_s(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r.a[i]) : "s"(sk), "v"(r.c));
#define S_pkfma(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r.p[i]) : "v"(r.pb), "v"(r.pc));
#define S_pkadd(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r.p[i]) : "v"(r.pb));
#define S_pkmul(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r.p[i]) : "v"(r.pb));
#define S_med3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r.a[i]) : "v"(r.b), "v"(r.c));
#define S_add(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r.a[i]) : "v"(r.b));
#define S_sub(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r.a[i]) : "v"(r.b));
#define S_mul(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r.a[i]) : "v"(r.b));
#define S_max(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r.a[i]) : "v"(r.b));
#define S_min(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r.a[i]) : "v"(r.b));
#define S_subclamp(i) asm volatile("v_sub_f32_e64 %0, %0, %1 clamp" : "+v"(r.a[i]) : "v"(r.b));
#define S_and(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r.u[i]) : "v"(r.ub));
#define S_and_s(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r.u[i]) : "s"(sk));
#define S_and_lit(i) asm volatile("v_and_b32 %0, 0xC0000000, %0" : "+v"(r.u[i]));
#define S_or(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r.u[i]) : "v"(r.ub));
#define S_ashr(i) asm volatile("v_ashrrev_i32 %0, 30, %0" : "+v"(r.u[i]));
#define S_lshl(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r.u[i]));
#define S_lshr(i) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(r.u[i]));
#define S_addu(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r.u[i]) : "v"(r.ub));
#define S_subu(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r.u[i]) : "v"(r.ub));
#define S_mul24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r.u[i]) : "v"(r.ub));
#define S_mad24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r.u[i]) : "v"(r.ub), "v"(r.uc));
#define S_mov(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r.u[i]) : "v"(r.ub));
#define S_cvtfi(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(r.a[i]) : "v"(r.u[i]));
#define S_cvtif(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(r.u[i]) : "v"(r.a[i]));
#define S_bfe(i) asm volatile("v_bfe_u32 %0, %0, 3, 10" : "+v"(r.u[i]));
#define S_bfei(i) asm volatile("v_bfe_i32 %0, %0, 3, 10" : "+v"(r.u[i]));
#define S_add3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r.u[i]) : "v"(r.ub), "v"(r.uc));
#define S_lshladd(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r.u[i]) : "v"(r.ub));
#define S_lshlor(i) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(r.u[i]) : "v"(r.ub));
#define S_andor(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r.u[i]) : "v"(r.ub), "v"(r.uc));
#define S_perm(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r.u[i]) : "v"(r.ub), "v"(r.uc));
#define S_minu(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r.u[i]) : "v"(r.ub));
#define S_lshladd64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r.q[i]) : "v"(r.q[(i + 1) & 15]));
#define S_mad64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r.q[i]) : "v"(r.ub), "v"(r.uc) : "vcc");
#define S_fma64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r.d[i]) : "v"(r.db), "v"(r.dc));
#define S_add64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r.d[i]) : "v"(r.db));
#define S_mul64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r.d[i]) : "v"(r.db));
#define S_cvt64(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r.d[i]) : "v"(r.a[i]));
#define S_cvt32(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r.a[i]) : "v"(r.d[i]));
#define S_adddpp(i) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(r.a[i]) : "v"(r.b));
#define S_movdpp(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r.u[i]) : "v"(r.ub));
#define S_cmp(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(r.a[i]), "v"(r.b) : "vcc");
#define S_cndmask(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r.u[i]) : "v"(r.ub) : "vcc");
#define S_dot4(i) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(r.u[i]) : "v"(r.ub), "v"(r.uc));
#define S_dot2(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(r.a[i]) : "v"(r.ub), "v"(r.uc));
#define S_pknorm(i) asm volatile("v_cvt_pknorm_i16_f32 %0, %1, %2" : "=v"(r.u[i]) : "v"(r.a[i]), "v"(r.b));
#define S_pkrtz(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r.u[i]) : "v"(r.a[i]), "v"(r.b));
#define S_sad(i) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(r.u[i]) : "v"(r.ub), "v"(r.uc));
#define S_max3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r.a[i]) : "v"(r.b), "v"(r.c));
#define S_xor(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r.u[i]) : "v"(r.ub));
#define S_bfi(i) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(r.u[i]) : "v"(r.ub), "v"(r.uc));
#define S_ldexp(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r.a[i]) : "v"(r.uc));
#define S_pkmov(i) asm volatile("v_pk_mov_b32 %0, %1, %2" : "=v"(r.p[i]) : "v"(r.pb), "v"(r.pc));

#define ALL(X)                                                                                                 \
    X(fma) X(fmac) X(fma_s) X(pkfma) X(pkadd) X(pkmul) X(med3) X(add) X(sub) X(mul) X(max) X(min) X(subclamp)  \
    X(and) X(and_s) X(and_lit) X(or) X(ashr) X(lshl) X(lshr) X(addu) X(subu) X(mul24) X(mad24) X(mov)          \
    X(cvtfi) X(cvtif) X(bfe) X(bfei) X(add3) X(lshladd) X(lshlor) X(andor) X(perm) X(minu) X(lshladd64)       \
    X(mad64) X(fma64) X(add64) X(mul64) X(cvt64) X(cvt32) X(adddpp) X(movdpp) X(cmp) X(cndmask) X(dot4)       \
    X(dot2) X(pknorm) X(pkrtz) X(sad) X(max3) X(xor) X(bfi) X(ldexp) X(pkmov)

#define DEF(n) KERNEL(k_##n, S_##n)
ALL(DEF)

typedef void (*kern_t)(float*);
struct Entry
{
    const char* name;
    kern_t fn;
};
#define ENT(n) {#n, k_##n},
static Entry entries[] = {ALL(ENT)};

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%d CUs; cycles at a nominal 2.4 GHz, per SIMD and wave-instruction\n", cus);
    float* out;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-12s %9s %9s %9s %9s\n", "instruction", "1 w/SIMD", "2", "4", "8");
    for (const Entry& e : entries)
    {
        printf("%-12s", e.name);
        for (int bpc : {1, 2, 4, 8})
        {
            const int blocks = cus * bpc;
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %9.2f", ms * 1e-3 * 2.4e9 / ((double)REPS * 64 * bpc));
        }
        printf("\n");
    }
    return 0;
}
