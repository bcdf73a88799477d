// Built-in TV-L1 dense optical flow for gfx950.
#include "srx_tvl1.h"

#include <algorithm>

namespace srx
{

namespace
{
__global__ void k_split2(const float2* __restrict__ uv, size_t count, float* __restrict__ u, float* __restrict__ v)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count)
        return;
    const float2 f = uv[e];
    u[e] = f.x;
    v[e] = f.y;
}

__global__ void k_merge2(const float* __restrict__ u, const float* __restrict__ v, size_t count, float2* __restrict__ uv)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count)
        uv[e] = make_float2(u[e], v[e]);
}

// plain streaming copy, 16 bytes per lane: what the memory system gives a kernel that only moves bytes
// One 16-byte element per thread, streaming loads and stores: of the plain copies tried on MI355X (grid-stride
// loops, 2 to 8 elements per thread, with and without the streaming hint; scripts/ubench_copy.hip) this is the
// fastest, 6.5 TB/s read + write on 1 GiB against 4.6 TB/s for a grid-stride loop over 8192 blocks.
typedef float copy_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy16(const copy_f4* __restrict__ src, copy_f4* __restrict__ dst, size_t count)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e < count)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + e), dst + e);
}

int bad_name(const std::string& n)
{
    set_error("unknown flow parameter: " + n);
    return SRX_ERR_INVALID;
}
} // namespace

int launch_split2(const float* uv, size_t count, float* u, float* v, hipStream_t st)
{
    const int bs = 256;
    hipLaunchKernelGGL(k_split2, dim3((unsigned)((count + bs - 1) / bs)), dim3(bs), 0, st, (const float2*)uv, count, u,
                       v);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "split2");
}

int launch_merge2(const float* u, const float* v, size_t count, float* uv, hipStream_t st)
{
    const int bs = 256;
    hipLaunchKernelGGL(k_merge2, dim3((unsigned)((count + bs - 1) / bs)), dim3(bs), 0, st, u, v, count, (float2*)uv);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "merge2");
}

int launch_copy16(const void* src, void* dst, size_t bytes, hipStream_t st)
{
    const size_t count = bytes / 16;
    if (count == 0)
        return 0;
    if ((count + 255) / 256 > 0x7FFFFFFFu)
    {
        set_error("copy16: more than 2^31 blocks");
        return SRX_ERR_INVALID;
    }
    const unsigned blocks = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, st, (const copy_f4*)src, (copy_f4*)dst, count);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "copy16");
}

int tvl1_set_int(Tvl1Params& p, const std::string& n, int v)
{
    if (n == "iterationsPerLaunch")
    {
        if (v != 0 && v != 1 && v != 4 && v != 8 && v != 12 && v != 14)
        {
            set_error("flow parameter iterationsPerLaunch must be 0 (by level size), 1, 4 or 8, or 12 / 14 (2 / 4 iterations "
                      "per launch on whole-line regions)");
            return SRX_ERR_INVALID;
        }
        p.iterationsPerLaunch = v;
        return SRX_OK;
    }
    if (v < 1)
    {
        set_error("flow parameter " + n + " must be positive");
        return SRX_ERR_INVALID;
    }
    if (n == "nscales")
        p.nscales = v;
    else if (n == "warps")
        p.warps = v;
    else if (n == "iterations")
        p.iterations = v;
    else
        return bad_name(n);
    return SRX_OK;
}

int tvl1_set_double(Tvl1Params& p, const std::string& n, double v)
{
    if (!(v > 0.0))
    {
        set_error("flow parameter " + n + " must be positive");
        return SRX_ERR_INVALID;
    }
    if (n == "tau")
        p.tau = v;
    else if (n == "lambda")
        p.lambda = v;
    else if (n == "theta")
        p.theta = v;
    else if (n == "epsilon")
        p.epsilon = v;
    else
        return bad_name(n);
    return SRX_OK;
}

int tvl1_get_int(const Tvl1Params& p, const std::string& n, int* v)
{
    if (!v)
        return bad_name(n);
    if (n == "nscales")
        *v = p.nscales;
    else if (n == "warps")
        *v = p.warps;
    else if (n == "iterations")
        *v = p.iterations;
    else if (n == "iterationsPerLaunch")
        *v = p.iterationsPerLaunch;
    else
        return bad_name(n);
    return SRX_OK;
}

int tvl1_get_double(const Tvl1Params& p, const std::string& n, double* v)
{
    if (!v)
        return bad_name(n);
    if (n == "tau")
        *v = p.tau;
    else if (n == "lambda")
        *v = p.lambda;
    else if (n == "theta")
        *v = p.theta;
    else if (n == "epsilon")
        *v = p.epsilon;
    else
        return bad_name(n);
    return SRX_OK;
}

// ---------------------------------------------------------------------------
// Kernels.  Every float operation follows the order of the CPU statement of this flow
// (oracle/tvl1_oracle.c in this repository's tests); the file is built with -ffp-contract=off.
namespace
{

__device__ __forceinline__ int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len)
        return p;
    if (len == 1)
        return 0;
    do
    {
        p = p < 0 ? -p : 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

// frame -> 8-bit grey as convertToType(frame, CV_8U, 1) does (lib/src/input_array_utility.cpp:77-124), as float
// kind: 0 = 8-bit, 1 = float (x 255), 2 = 16-bit (x 255/65535 in float, as cv::Mat::convertTo scales)
__global__ void k_to_gray(const void* __restrict__ src, size_t count, int ch, int kind, float* __restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    int c[4] = {0, 0, 0, 0};
    for (int k = 0; k < ch; ++k)
    {
        if (kind != 0)
        {
            const float v = kind == 1 ? ((const float*)src)[i * ch + k] * 255.0f
                                      : (float)((const uint16_t*)src)[i * ch + k] * (float)(255.0 / 65535.0);
            float r = rintf(v);
            r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
            c[k] = (v != v) ? 0 : (int)r;
        }
        else
            c[k] = ((const uint8_t*)src)[i * ch + k];
    }
    dst[i] = ch == 1 ? (float)c[0] : (float)((c[0] * 1868 + c[1] * 9617 + c[2] * 4899 + 8192) >> 14);
}

// 5-tap binomial {1,4,6,4,1}, REFLECT_101, even samples
__global__ void k_pyr_down(const float* __restrict__ src0, const float* __restrict__ src1, int rows, int cols,
                           float* __restrict__ dst0, float* __restrict__ dst1, int dr, int dc)
{
    // both images of the pair in one launch: blockIdx.z picks the image
    const float* __restrict__ src = blockIdx.z ? src1 : src0;
    float* __restrict__ dst = blockIdx.z ? dst1 : dst0;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dc || y >= dr)
        return;
    const int c0 = reflect101(2 * x - 2, cols), c1 = reflect101(2 * x - 1, cols), c2 = 2 * x;
    const int c3 = reflect101(2 * x + 1, cols), c4 = reflect101(2 * x + 2, cols);
    float r[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
    {
        const float* S = src + (size_t)reflect101(2 * y - 2 + j, rows) * cols;
        r[j] = S[c2] * 6 + (S[c1] + S[c3]) * 4 + S[c0] + S[c4];
    }
    dst[(size_t)y * dc + x] = (r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4]) * 0.00390625f;
}

// All launches of a frame pair are queued without the host looking at anything in between.  The pair is a chain
// of phases, one per (level, warp); *stop is the first phase that must not run yet because the phase before it
// has not finished within the launches queued for it (k_boundary).  Every kernel of phase `phase` and later
// then returns at once, which leaves the unfinished phase's state untouched for the host to continue from.
struct Guard
{
    const int* stop;
    int phase;
};
__device__ __forceinline__ bool stopped(const Guard g)
{
    return g.stop != nullptr && *g.stop <= g.phase;
}

__device__ __forceinline__ void cubic_coeffs(float x, float (&c)[4])
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

// bicubic warp of I1, I1x, I1y by (u1, u2) = (a, b) at pixel (x, y), zero outside; grad = |grad I1w|^2,
// rho_c = I1w - I1wx*u1 - I1wy*u2 - I0
__device__ __forceinline__ void warp_at(const float* __restrict__ I0, const float* __restrict__ I1,
                                        const float* __restrict__ I1x, const float* __restrict__ I1y, float a, float b,
                                        int x, int y, int rows, int cols, float* __restrict__ I1wx,
                                        float* __restrict__ I1wy, float* __restrict__ grad, float* __restrict__ rho_c)
{
    const size_t i = (size_t)y * cols + x;
    const float fx = (float)x + a, fy = (float)y + b;
    float w = 0.f, wx = 0.f, wy = 0.f;
    if (fabsf(fx) < 1e8f && fabsf(fy) < 1e8f)
    {
        const float flx = floorf(fx), fly = floorf(fy);
        const int sx = (int)flx, sy = (int)fly;
        float cx[4], cy[4];
        cubic_coeffs(fx - flx, cx);
        cubic_coeffs(fy - fly, cy);
        float h0[4], h1[4], h2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const int yy = sy - 1 + j;
            float t0[4], t1[4], t2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const int xx = sx - 1 + k;
                const bool in = (unsigned)xx < (unsigned)cols && (unsigned)yy < (unsigned)rows;
                const size_t o = in ? (size_t)yy * cols + xx : 0;
                t0[k] = in ? I1[o] : 0.f;
                t1[k] = in ? I1x[o] : 0.f;
                t2[k] = in ? I1y[o] : 0.f;
            }
            h0[j] = t0[0] * cx[0] + t0[1] * cx[1] + t0[2] * cx[2] + t0[3] * cx[3];
            h1[j] = t1[0] * cx[0] + t1[1] * cx[1] + t1[2] * cx[2] + t1[3] * cx[3];
            h2[j] = t2[0] * cx[0] + t2[1] * cx[1] + t2[2] * cx[2] + t2[3] * cx[3];
        }
        w = h0[0] * cy[0] + h0[1] * cy[1] + h0[2] * cy[2] + h0[3] * cy[3];
        wx = h1[0] * cy[0] + h1[1] * cy[1] + h1[2] * cy[2] + h1[3] * cy[3];
        wy = h2[0] * cy[0] + h2[1] * cy[1] + h2[2] * cy[2] + h2[3] * cy[3];
    }
    I1wx[i] = wx;
    I1wy[i] = wy;
    grad[i] = wx * wx + wy * wy;
    rho_c[i] = w - wx * a - wy * b - I0[i];
}

__global__ void k_warp(const float* __restrict__ I0, const float* __restrict__ I1, const float* __restrict__ I1x,
                       const float* __restrict__ I1y, const float* __restrict__ u1, const float* __restrict__ u2,
                       int rows, int cols, float* __restrict__ I1wx, float* __restrict__ I1wy,
                       float* __restrict__ grad, float* __restrict__ rho_c, Guard guard)
{
    if (stopped(guard))
        return;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows)
        return;
    const size_t i = (size_t)y * cols + x;
    warp_at(I0, I1, I1x, I1y, u1[i], u2[i], x, y, rows, cols, I1wx, I1wy, grad, rho_c);
}

__device__ __forceinline__ unsigned long long error_term(float du1, float du2)
{
    float q = du1 * du1 + du2 * du2;
    if (!(q < 4096.0f))
        q = 4096.0f;
    return (unsigned long long)(q * 1073741824.0f);
}

// One inner iteration in one launch: thresholding step, divergence of p, primal update and error,
// then the forward gradient of the new u and the dual update.
//
// The dual update at a pixel needs the new u of its right and lower neighbours, so a block computes
// the new u for its 32x8 tile plus one column to the right and one row below (those are recomputed,
// not exchanged) and keeps them in LDS.  Because a neighbouring block recomputes pixels of this tile
// from the old values, nothing is updated in place: u and p are read from one buffer set and
// written to the other, and the host swaps the sets every iteration.
//
// err[it] receives the iteration's fixed-point error (integer adds: order independent).  The
// iteration runs only if the previous one's error exceeded the threshold; err[] starts at zero, so
// once one iteration has been skipped or has converged all later ones see "converged" too (and the
// result stays in the set the last executed iteration wrote).
struct Tvl1Planes
{
    const float *I1wx, *I1wy, *grad, *rho_c;
    const float *u1, *u2, *p11, *p12, *p21, *p22; // read
    float *nu1, *nu2, *np11, *np12, *np21, *np22; // written
};

constexpr int kItX = 32, kItY = 8;

__device__ __forceinline__ void tvl1_new_u(const Tvl1Planes& P, int x, int y, int cols, float l_t, float theta,
                                           float& n1, float& n2, float& a, float& b)
{
    const size_t i = (size_t)y * cols + x;
    const float gx = P.I1wx[i], gy = P.I1wy[i], g = P.grad[i];
    a = P.u1[i], b = P.u2[i];
    const float rho = P.rho_c[i] + (gx * a + gy * b);
    float d1 = 0.0f, d2 = 0.0f;
    if (rho < -l_t * g)
    {
        d1 = l_t * gx;
        d2 = l_t * gy;
    }
    else if (rho > l_t * g)
    {
        d1 = -l_t * gx;
        d2 = -l_t * gy;
    }
    else if (g > 1.1920929e-07f)
    {
        const float fi = -rho / g;
        d1 = fi * gx;
        d2 = fi * gy;
    }
    const float v1 = a + d1, v2 = b + d2;
    const float q11 = P.p11[i], q12 = P.p12[i], q21 = P.p21[i], q22 = P.p22[i];
    const float div1 = (x > 0 ? q11 - P.p11[i - 1] : q11) + (y > 0 ? q12 - P.p12[i - cols] : q12);
    const float div2 = (x > 0 ? q21 - P.p21[i - 1] : q21) + (y > 0 ? q22 - P.p22[i - cols] : q22);
    n1 = v1 + theta * div1;
    n2 = v2 + theta * div2;
}

__global__ __launch_bounds__(256) void k_tvl1_iter(Tvl1Planes P, int rows, int cols, float l_t, float theta, float taut,
                                                    unsigned long long* __restrict__ err, int it,
                                                    unsigned long long thr, Guard guard)
{
    if (stopped(guard) || (it > 0 && err[it - 1] <= thr))
        return;
    __shared__ float su1[(kItY + 1) * (kItX + 1)], su2[(kItY + 1) * (kItX + 1)];
    __shared__ unsigned long long part[4];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * kItX + tx;
    const int x0 = blockIdx.x * kItX, y0 = blockIdx.y * kItY;
    const int x = x0 + tx, y = y0 + ty;
    const bool own = x < cols && y < rows;
    unsigned long long e = 0;
    float n1 = 0.f, n2 = 0.f;
    if (own)
    {
        float a, b;
        tvl1_new_u(P, x, y, cols, l_t, theta, n1, n2, a, b);
        e = error_term(n1 - a, n2 - b);
        su1[ty * (kItX + 1) + tx] = n1;
        su2[ty * (kItX + 1) + tx] = n2;
    }
    // the column right of the tile (threads 0..7) and the row below it (threads 8..39)
    if (tid < kItY + kItX)
    {
        const int hx = tid < kItY ? x0 + kItX : x0 + (tid - kItY);
        const int hy = tid < kItY ? y0 + tid : y0 + kItY;
        if (hx < cols && hy < rows)
        {
            float h1, h2, a, b;
            tvl1_new_u(P, hx, hy, cols, l_t, theta, h1, h2, a, b);
            su1[(hy - y0) * (kItX + 1) + (hx - x0)] = h1;
            su2[(hy - y0) * (kItX + 1) + (hx - x0)] = h2;
        }
    }
    __syncthreads();
    if (own)
    {
        const size_t i = (size_t)y * cols + x;
        const int l = ty * (kItX + 1) + tx;
        const float u1x = x < cols - 1 ? su1[l + 1] - n1 : 0.0f, u1y = y < rows - 1 ? su1[l + kItX + 1] - n1 : 0.0f;
        const float u2x = x < cols - 1 ? su2[l + 1] - n2 : 0.0f, u2y = y < rows - 1 ? su2[l + kItX + 1] - n2 : 0.0f;
        const float g1 = sqrtf(u1x * u1x + u1y * u1y), g2 = sqrtf(u2x * u2x + u2y * u2y);
        const float ng1 = 1.0f + taut * g1, ng2 = 1.0f + taut * g2;
        P.nu1[i] = n1;
        P.nu2[i] = n2;
        P.np11[i] = (P.p11[i] + taut * u1x) / ng1;
        P.np12[i] = (P.p12[i] + taut * u1y) / ng1;
        P.np21[i] = (P.p21[i] + taut * u2x) / ng2;
        P.np22[i] = (P.p22[i] + taut * u2y) / ng2;
    }
    // block reduction of the integer error, one atomic per block
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        e += __shfl_down(e, o, 64);
    if ((tid & 63) == 0)
        part[tid >> 6] = e;
    __syncthreads();
    if (tid == 0)
        atomicAdd(&err[it], part[0] + part[1] + part[2] + part[3]);
}

// The same iteration with four pixels of a row per thread and 16-byte accesses, for levels whose
// width is a multiple of 4 (every plane is then 16-byte aligned row by row).  A block covers
// 128 x kIt4Y pixels.  Values per pixel are computed exactly as in k_tvl1_iter.
// rows of a tile (even; 12 measured best of 8 .. 30 at 1920x1080: the halo row is a twelfth of the loads, and
// 1080, 540 and 270 rows are whole numbers of tiles)
#ifndef SRX_IT4_ROWS
#define SRX_IT4_ROWS 12
#endif
constexpr int kIt4X = 128, kIt4Y = SRX_IT4_ROWS, kIt4Pitch = kIt4X + 4;

__device__ __forceinline__ void tvl1_u_values(float gx, float gy, float g, float rc, float a, float b, float q11,
                                              float q12, float q21, float q22, float left11, float left21, float up12,
                                              float up22, bool has_left, bool has_up, float l_t, float theta,
                                              float& n1, float& n2)
{
    const float rho = rc + (gx * a + gy * b);
    float d1 = 0.0f, d2 = 0.0f;
    if (rho < -l_t * g)
    {
        d1 = l_t * gx;
        d2 = l_t * gy;
    }
    else if (rho > l_t * g)
    {
        d1 = -l_t * gx;
        d2 = -l_t * gy;
    }
    else if (g > 1.1920929e-07f)
    {
        const float fi = -rho / g;
        d1 = fi * gx;
        d2 = fi * gy;
    }
    const float v1 = a + d1, v2 = b + d2;
    const float div1 = (has_left ? q11 - left11 : q11) + (has_up ? q12 - up12 : q12);
    const float div2 = (has_left ? q21 - left21 : q21) + (has_up ? q22 - up22 : q22);
    n1 = v1 + theta * div1;
    n2 = v2 + theta * div2;
}

__global__ __launch_bounds__(32 * (kIt4Y + 2)) void k_tvl1_iter4(Tvl1Planes P, int rows, int cols, float l_t, float theta,
                                                     float taut, unsigned long long* __restrict__ err, int it,
                                                     unsigned long long thr, Guard guard)
{
    if (stopped(guard) || (it > 0 && err[it - 1] <= thr))
        return;
    __shared__ __attribute__((aligned(16))) float su1[(kIt4Y + 1) * kIt4Pitch], su2[(kIt4Y + 1) * kIt4Pitch];
    __shared__ unsigned long long part[(kIt4Y + 2) / 2];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 32 + tx;
    const int x0 = blockIdx.x * kIt4X, y0 = blockIdx.y * kIt4Y;

    // new u of four pixels starting at (x, y); returns false outside the image
    auto new_u4 = [&](int x, int y, float (&n1)[4], float (&n2)[4], float (&a)[4], float (&b)[4], float4 (&q)[4]) {
        if (x >= cols || y >= rows)
            return false;
        const size_t i = (size_t)y * cols + x;
        const float4 gx = *(const float4*)(P.I1wx + i), gy = *(const float4*)(P.I1wy + i);
        const float4 g = *(const float4*)(P.grad + i), rc = *(const float4*)(P.rho_c + i);
        const float4 u1 = *(const float4*)(P.u1 + i), u2 = *(const float4*)(P.u2 + i);
        q[0] = *(const float4*)(P.p11 + i), q[1] = *(const float4*)(P.p12 + i);
        q[2] = *(const float4*)(P.p21 + i), q[3] = *(const float4*)(P.p22 + i);
        float4 up12 = make_float4(0.f, 0.f, 0.f, 0.f), up22 = up12;
        if (y > 0)
        {
            up12 = *(const float4*)(P.p12 + i - cols);
            up22 = *(const float4*)(P.p22 + i - cols);
        }
        float l11 = 0.f, l21 = 0.f;
        if (x > 0)
        {
            l11 = P.p11[i - 1];
            l21 = P.p21[i - 1];
        }
        const float gxv[4] = {gx.x, gx.y, gx.z, gx.w}, gyv[4] = {gy.x, gy.y, gy.z, gy.w};
        const float gv[4] = {g.x, g.y, g.z, g.w}, rcv[4] = {rc.x, rc.y, rc.z, rc.w};
        const float q11[4] = {q[0].x, q[0].y, q[0].z, q[0].w}, q12[4] = {q[1].x, q[1].y, q[1].z, q[1].w};
        const float q21[4] = {q[2].x, q[2].y, q[2].z, q[2].w}, q22[4] = {q[3].x, q[3].y, q[3].z, q[3].w};
        const float u12[4] = {up12.x, up12.y, up12.z, up12.w}, u22[4] = {up22.x, up22.y, up22.z, up22.w};
        a[0] = u1.x, a[1] = u1.y, a[2] = u1.z, a[3] = u1.w;
        b[0] = u2.x, b[1] = u2.y, b[2] = u2.z, b[3] = u2.w;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            tvl1_u_values(gxv[e], gyv[e], gv[e], rcv[e], a[e], b[e], q11[e], q12[e], q21[e], q22[e],
                          e == 0 ? l11 : q11[e - 1], e == 0 ? l21 : q21[e - 1], u12[e], u22[e], x + e > 0, y > 0, l_t,
                          theta, n1[e], n2[e]);
        return true;
    };

    // Rows 0 .. kIt4Y-1 of the block's threads own the tile's pixels; row kIt4Y works out the new u of the row
    // below the tile and the first kIt4Y threads of row kIt4Y+1 that of the column right of it (four pixels each,
    // the first is used), which the tile's last row and column need for their forward gradients.  Those two rows
    // are a wave of their own: every wave makes one pass through new_u4 before the barrier.
    const bool halo_row = ty == kIt4Y, halo_col = ty == kIt4Y + 1;
    const int x = halo_col ? x0 + kIt4X : x0 + 4 * tx, y = halo_col ? y0 + tx : y0 + ty;
    float n1[4], n2[4], a[4], b[4];
    float4 q[4];
    unsigned long long e = 0;
    const bool have = !(halo_col && tx >= kIt4Y) && new_u4(x, y, n1, n2, a, b, q);
    const bool own = have && ty < kIt4Y;
    if (own)
    {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            e += error_term(n1[k] - a[k], n2[k] - b[k]);
    }
    if (have && !halo_col)
    {
        *(float4*)&su1[ty * kIt4Pitch + 4 * tx] = make_float4(n1[0], n1[1], n1[2], n1[3]);
        *(float4*)&su2[ty * kIt4Pitch + 4 * tx] = make_float4(n2[0], n2[1], n2[2], n2[3]);
    }
    else if (have)
    {
        su1[tx * kIt4Pitch + kIt4X] = n1[0];
        su2[tx * kIt4Pitch + kIt4X] = n2[0];
    }
    (void)halo_row;
    __syncthreads();
    if (own)
    {
        const size_t i = (size_t)y * cols + x;
        const int l = ty * kIt4Pitch + 4 * tx;
        float o11[4], o12[4], o21[4], o22[4];
        const float q11[4] = {q[0].x, q[0].y, q[0].z, q[0].w}, q12[4] = {q[1].x, q[1].y, q[1].z, q[1].w};
        const float q21[4] = {q[2].x, q[2].y, q[2].z, q[2].w}, q22[4] = {q[3].x, q[3].y, q[3].z, q[3].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            const bool has_r = x + k < cols - 1, has_d = y < rows - 1;
            const float r1 = k < 3 ? n1[k + 1] : su1[l + 4], r2 = k < 3 ? n2[k + 1] : su2[l + 4];
            const float u1x = has_r ? r1 - n1[k] : 0.0f, u1y = has_d ? su1[l + kIt4Pitch + k] - n1[k] : 0.0f;
            const float u2x = has_r ? r2 - n2[k] : 0.0f, u2y = has_d ? su2[l + kIt4Pitch + k] - n2[k] : 0.0f;
            const float g1 = sqrtf(u1x * u1x + u1y * u1y), g2 = sqrtf(u2x * u2x + u2y * u2y);
            const float ng1 = 1.0f + taut * g1, ng2 = 1.0f + taut * g2;
            o11[k] = (q11[k] + taut * u1x) / ng1;
            o12[k] = (q12[k] + taut * u1y) / ng1;
            o21[k] = (q21[k] + taut * u2x) / ng2;
            o22[k] = (q22[k] + taut * u2y) / ng2;
        }
        *(float4*)(P.nu1 + i) = make_float4(n1[0], n1[1], n1[2], n1[3]);
        *(float4*)(P.nu2 + i) = make_float4(n2[0], n2[1], n2[2], n2[3]);
        *(float4*)(P.np11 + i) = make_float4(o11[0], o11[1], o11[2], o11[3]);
        *(float4*)(P.np12 + i) = make_float4(o12[0], o12[1], o12[2], o12[3]);
        *(float4*)(P.np21 + i) = make_float4(o21[0], o21[1], o21[2], o21[3]);
        *(float4*)(P.np22 + i) = make_float4(o22[0], o22[1], o22[2], o22[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        e += __shfl_down(e, o, 64);
    if ((tid & 63) == 0)
        part[tid >> 6] = e;
    __syncthreads();
    if (tid == 0)
    {
        unsigned long long sum = 0;
#pragma unroll
        for (int wv = 0; wv < kIt4Y / 2; ++wv) // the waves that own pixels
            sum += part[wv];
        atomicAdd(&err[it], sum);
    }
    static_assert(kIt4Y % 2 == 0, "whole waves own pixels");
}

// Several inner iterations in one launch.  A block owns a tile of TW x TH pixels and loads it with a halo of
// KT pixels on every side; each iteration is carried out on the whole region held by the block, and what a
// pixel at the region's edge cannot know (its neighbour lies outside) makes the outermost ring of the region
// wrong by one pixel more per iteration - after KT iterations the tile itself is still exact.  Every value of
// the tile is computed by the same float operations, from the same operands, as by one launch per iteration
// (k_tvl1_iter); only pixels of the halo are computed more than once, by the blocks that share them.
//   state per pixel: u1, u2, p11, p12, p21, p22 and the four constants of the warp, in registers;
//   LDS: p11, p21 (read by the right neighbour), p12, p22 (by the lower one), new u1, u2 (by the left and
//   upper ones for the forward gradient) of the whole region.
// Launch b covers iterations it0 = b*KT .. it0+n_iter-1, reads buffer set b & 1 and writes the other one, and
// adds each iteration's error over its tile to err[it].  The iteration stops after the first iteration j whose
// error is at most thr (lib side: the published TV-L1 loop "while (error > epsilon^2 * area && n < iterations)").
// A launch therefore looks at the errors of all earlier iterations first:
//   none at or below thr: it runs;
//   the first such j is the last iteration of the launch before: done, the result is what that launch wrote
//   (the set this launch would read); it returns, as do all launches after it;
//   the first such j lies inside the launch before, not at its end: that launch ran too far.  Its input set
//   is still intact (this launch is the only one that would overwrite it), so this launch repeats the
//   iterations up to j from that set and writes the result over the other one - the same set as in the case
//   above - without touching err[];
//   j is older than that: it returns.
template <int KT, int TW, int TH, int NT>
__global__ __launch_bounds__(NT, NT == 256 ? 4 : 4) void k_tvl1_multi(Tvl1Planes P, int rows, int cols, float l_t, float theta,
                                                     float taut, unsigned long long* __restrict__ err, int n_err,
                                                     int it0, int n_iter, unsigned long long thr, Guard guard)
{
    if (stopped(guard))
        return;
    constexpr int RW = TW + 2 * KT, RH = TH + 2 * KT, CELLS = RW * RH, PER = (CELLS + NT - 1) / NT;
    __shared__ float s_p11[CELLS], s_p21[CELLS], s_p12[CELLS], s_p22[CELLS], s_u1[CELLS], s_u2[CELLS];
    __shared__ unsigned long long s_err[KT][NT / 64];
    __shared__ int s_first;
    const int tid = threadIdx.x;
    if (tid < 64)
    {
        int first = 0x7FFFFFFF;
        for (int j = tid; j < min(it0, n_err); j += 64) // err[] has n_err entries
            if (err[j] <= thr)
            {
                first = j;
                break;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            first = min(first, __shfl_down(first, o, 64));
        if (tid == 0)
            s_first = first;
    }
    __syncthreads();
    const int first = s_first;
    bool redo = false;
    if (first != 0x7FFFFFFF)
    {
        if (first < it0 - KT || first == it0 - 1)
            return;
        redo = true;
        n_iter = first - (it0 - KT) + 1;
    }
    // the sets as this launch uses them (a repeat reads what the launch before read)
    const float* ru1 = redo ? P.nu1 : P.u1;
    const float* ru2 = redo ? P.nu2 : P.u2;
    const float* rp11 = redo ? P.np11 : P.p11;
    const float* rp12 = redo ? P.np12 : P.p12;
    const float* rp21 = redo ? P.np21 : P.p21;
    const float* rp22 = redo ? P.np22 : P.p22;
    float* wu1 = redo ? const_cast<float*>(P.u1) : P.nu1;
    float* wu2 = redo ? const_cast<float*>(P.u2) : P.nu2;
    float* wp11 = redo ? const_cast<float*>(P.p11) : P.np11;
    float* wp12 = redo ? const_cast<float*>(P.p12) : P.np12;
    float* wp21 = redo ? const_cast<float*>(P.p21) : P.np21;
    float* wp22 = redo ? const_cast<float*>(P.p22) : P.np22;

    const int x0 = blockIdx.x * TW - KT, y0 = blockIdx.y * TH - KT; // image position of the region's first cell
    float cgx[PER], cgy[PER], cg[PER], crc[PER], a[PER], b[PER], q11[PER], q12[PER], q21[PER], q22[PER];
    bool inside[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i)
    {
        const int c = tid + NT * i;
        const int cy = c / RW, cx = c - cy * RW;
        const int gx = x0 + cx, gy = y0 + cy;
        inside[i] = c < CELLS && gx >= 0 && gx < cols && gy >= 0 && gy < rows;
        cgx[i] = cgy[i] = cg[i] = crc[i] = a[i] = b[i] = q11[i] = q12[i] = q21[i] = q22[i] = 0.0f;
        if (inside[i])
        {
            const size_t o = (size_t)gy * cols + gx;
            cgx[i] = P.I1wx[o], cgy[i] = P.I1wy[o], cg[i] = P.grad[o], crc[i] = P.rho_c[o];
            a[i] = ru1[o], b[i] = ru2[o];
            q11[i] = rp11[o], q12[i] = rp12[o], q21[i] = rp21[o], q22[i] = rp22[o];
        }
        if (c < CELLS)
            s_p11[c] = q11[i], s_p21[c] = q21[i], s_p12[c] = q12[i], s_p22[c] = q22[i];
    }
    unsigned long long e[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j)
        e[j] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KT; ++j)
    {
        if (j < n_iter) // the same for every thread of the block
        {
#pragma unroll
            for (int i = 0; i < PER; ++i)
            {
                const int c = tid + NT * i;
                if (!inside[i])
                    continue;
                const int cy = c / RW, cx = c - cy * RW;
                const int gx = x0 + cx, gy = y0 + cy;
                const float l11 = cx > 0 ? s_p11[c - 1] : 0.0f, l21 = cx > 0 ? s_p21[c - 1] : 0.0f;
                const float u12 = cy > 0 ? s_p12[c - RW] : 0.0f, u22 = cy > 0 ? s_p22[c - RW] : 0.0f;
                float n1, n2;
                tvl1_u_values(cgx[i], cgy[i], cg[i], crc[i], a[i], b[i], q11[i], q12[i], q21[i], q22[i], l11, l21, u12,
                              u22, gx > 0, gy > 0, l_t, theta, n1, n2);
                if (cx >= KT && cx < KT + TW && cy >= KT && cy < KT + TH)
                    e[j] += error_term(n1 - a[i], n2 - b[i]);
                a[i] = n1, b[i] = n2;
                s_u1[c] = n1, s_u2[c] = n2;
            }
        }
        __syncthreads();
        if (j < n_iter)
        {
#pragma unroll
            for (int i = 0; i < PER; ++i)
            {
                const int c = tid + NT * i;
                if (!inside[i])
                    continue;
                const int cy = c / RW, cx = c - cy * RW;
                const int gx = x0 + cx, gy = y0 + cy;
                const bool has_r = gx < cols - 1, has_d = gy < rows - 1;
                const float n1 = a[i], n2 = b[i];
                const float r1 = cx + 1 < RW ? s_u1[c + 1] : 0.0f, r2 = cx + 1 < RW ? s_u2[c + 1] : 0.0f;
                const float d1 = cy + 1 < RH ? s_u1[c + RW] : 0.0f, d2 = cy + 1 < RH ? s_u2[c + RW] : 0.0f;
                const float u1x = has_r ? r1 - n1 : 0.0f, u1y = has_d ? d1 - n1 : 0.0f;
                const float u2x = has_r ? r2 - n2 : 0.0f, u2y = has_d ? d2 - n2 : 0.0f;
                const float g1 = sqrtf(u1x * u1x + u1y * u1y), g2 = sqrtf(u2x * u2x + u2y * u2y);
                const float ng1 = 1.0f + taut * g1, ng2 = 1.0f + taut * g2;
                q11[i] = (q11[i] + taut * u1x) / ng1;
                q12[i] = (q12[i] + taut * u1y) / ng1;
                q21[i] = (q21[i] + taut * u2x) / ng2;
                q22[i] = (q22[i] + taut * u2y) / ng2;
                s_p11[c] = q11[i], s_p21[c] = q21[i], s_p12[c] = q12[i], s_p22[c] = q22[i];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < PER; ++i)
    {
        const int c = tid + NT * i;
        const int cy = c / RW, cx = c - cy * RW;
        if (inside[i] && cx >= KT && cx < KT + TW && cy >= KT && cy < KT + TH)
        {
            const size_t o = (size_t)(y0 + cy) * cols + (x0 + cx);
            wu1[o] = a[i], wu2[o] = b[i];
            wp11[o] = q11[i], wp12[o] = q12[i], wp21[o] = q21[i], wp22[o] = q22[i];
        }
    }
    if (redo)
        return;
#pragma unroll
    for (int j = 0; j < KT; ++j)
    {
        unsigned long long v = e[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            v += __shfl_down(v, o, 64);
        if ((tid & 63) == 0)
            s_err[j][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < KT && tid < n_iter)
    {
        unsigned long long sum = 0;
#pragma unroll
        for (int wv = 0; wv < NT / 64; ++wv)
            sum += s_err[tid][wv];
        atomicAdd(&err[it0 + tid], sum);
    }
}

// value held by lane-1 / lane+1 of the wave (0 at the wave's ends)
__device__ __forceinline__ float lane_left(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_right(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, true));
}

// Several inner iterations in one launch on regions whose rows are whole 512-byte runs of every plane: the form
// of k_tvl1_multi for levels whose width is a multiple of 4 (all planes are then 16-byte aligned row by row).
//   region: kRowsW = 128 columns (a tile of kRowsTW = 120 columns with 4 columns of halo on either side, so that
//   a region starts at a multiple of 4 floats) x RR = (NT / 32) * RPT rows (a tile of RR - 2 KT rows with KT rows
//   of halo above and below);
//   a thread owns four neighbouring pixels of RPT neighbouring rows and keeps their state (the warp's four
//   constants, u, p) in registers for the whole launch: one 16-byte load per plane, row and thread going in,
//   one 16-byte store per plane coming out;
//   neighbours: the pixels left and right of a thread's four are in the neighbouring lanes (a row is 32 lanes;
//   what the first and last lane of a row get from the other row of their wave only ever reaches the region's
//   outermost ring), the rows above and below a thread's rows go through LDS, one row per thread and direction.
// As in k_tvl1_multi every iteration makes the outermost ring of the region wrong by one more pixel and leaves
// the tile exact after KT iterations; every value of the tile is computed by the same operations from the same
// operands as by one launch per iteration.  The launch protocol (which iterations a launch covers, what it does
// once an earlier iteration has met the threshold) is that of k_tvl1_multi.
constexpr int kRowsW = 128, kRowsTW = 120, kRowsHX = 4;
// waves per SIMD the 4-iteration form is compiled for: at 4 (128 registers) it spills 15 registers, at 3 it needs
// 140 and none; measured 2-3 % faster per pair without the spills
#ifndef SRX_ROWS_WAVES
#define SRX_ROWS_WAVES 3
#endif
template <int KT, int RPT, int NT>
__global__ __launch_bounds__(NT, KT >= 4 ? SRX_ROWS_WAVES : 4) void k_tvl1_rows(Tvl1Planes P, int rows, int cols, float l_t, float theta, float taut,
                                                   unsigned long long* __restrict__ err, int n_err, int it0, int n_iter,
                                                   unsigned long long thr, Guard guard)
{
    if (stopped(guard))
        return;
    constexpr int TR = NT / 32;       // rows of threads
    constexpr int RR = TR * RPT;      // rows of the region
    constexpr int TH = RR - 2 * KT;   // rows of the tile
    static_assert(KT <= kRowsHX && TH > 0 && NT % 64 == 0, "halo and block shape");
    // what the thread row below / above needs of this thread row: p12, p22 of its last row, the new u of its first
    __shared__ __attribute__((aligned(16))) float s_p12[TR * kRowsW], s_p22[TR * kRowsW], s_n1[TR * kRowsW], s_n2[TR * kRowsW];
    __shared__ unsigned long long s_err[KT][NT / 64];
    __shared__ int s_first;
    const int tid = threadIdx.x;
    if (tid < 64)
    {
        int first = 0x7FFFFFFF;
        for (int j = tid; j < min(it0, n_err); j += 64) // err[] has n_err entries
            if (err[j] <= thr)
            {
                first = j;
                break;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            first = min(first, __shfl_down(first, o, 64));
        if (tid == 0)
            s_first = first;
    }
    __syncthreads();
    const int first = s_first;
    bool redo = false;
    if (first != 0x7FFFFFFF)
    {
        if (first < it0 - KT || first == it0 - 1)
            return;
        redo = true;
        n_iter = first - (it0 - KT) + 1;
    }
    const float* ru1 = redo ? P.nu1 : P.u1;
    const float* ru2 = redo ? P.nu2 : P.u2;
    const float* rp11 = redo ? P.np11 : P.p11;
    const float* rp12 = redo ? P.np12 : P.p12;
    const float* rp21 = redo ? P.np21 : P.p21;
    const float* rp22 = redo ? P.np22 : P.p22;
    float* wu1 = redo ? const_cast<float*>(P.u1) : P.nu1;
    float* wu2 = redo ? const_cast<float*>(P.u2) : P.nu2;
    float* wp11 = redo ? const_cast<float*>(P.p11) : P.np11;
    float* wp12 = redo ? const_cast<float*>(P.p12) : P.np12;
    float* wp21 = redo ? const_cast<float*>(P.p21) : P.np21;
    float* wp22 = redo ? const_cast<float*>(P.p22) : P.np22;

    const int lx = tid & 31, tr = tid >> 5;
    const int gx0 = (int)blockIdx.x * kRowsTW - kRowsHX + 4 * lx; // image column of the thread's first pixel
    const int gy0 = (int)blockIdx.y * TH - KT + tr * RPT;         // image row of the thread's first row
    const bool col_in = gx0 >= 0 && gx0 < cols;                   // cols % 4 == 0: four pixels are in or out together
    float cgx[RPT][4], cgy[RPT][4], cg[RPT][4], crc[RPT][4], a[RPT][4], b[RPT][4];
    float q11[RPT][4], q12[RPT][4], q21[RPT][4], q22[RPT][4];
    bool inside[RPT];
    auto spread = [](const float4 v, float (&d)[4]) { d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w; };
#pragma unroll
    for (int i = 0; i < RPT; ++i)
    {
        const int gy = gy0 + i;
        inside[i] = col_in && gy >= 0 && gy < rows;
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v[10] = {z, z, z, z, z, z, z, z, z, z};
        if (inside[i])
        {
            const size_t o = (size_t)gy * cols + gx0;
            v[0] = *(const float4*)(P.I1wx + o), v[1] = *(const float4*)(P.I1wy + o);
            v[2] = *(const float4*)(P.grad + o), v[3] = *(const float4*)(P.rho_c + o);
            v[4] = *(const float4*)(ru1 + o), v[5] = *(const float4*)(ru2 + o);
            v[6] = *(const float4*)(rp11 + o), v[7] = *(const float4*)(rp12 + o);
            v[8] = *(const float4*)(rp21 + o), v[9] = *(const float4*)(rp22 + o);
        }
        spread(v[0], cgx[i]), spread(v[1], cgy[i]), spread(v[2], cg[i]), spread(v[3], crc[i]);
        spread(v[4], a[i]), spread(v[5], b[i]);
        spread(v[6], q11[i]), spread(v[7], q12[i]), spread(v[8], q21[i]), spread(v[9], q22[i]);
    }
    const int srow = tr * kRowsW + 4 * lx; // this thread's place in the exchange rows
    *(float4*)&s_p12[srow] = make_float4(q12[RPT - 1][0], q12[RPT - 1][1], q12[RPT - 1][2], q12[RPT - 1][3]);
    *(float4*)&s_p22[srow] = make_float4(q22[RPT - 1][0], q22[RPT - 1][1], q22[RPT - 1][2], q22[RPT - 1][3]);
    unsigned long long e[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j)
        e[j] = 0;
    // which of the thread's pixels belong to the tile (the error is summed over tiles, the stores cover them)
    const bool tile_col = lx >= 1 && lx <= 30;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KT; ++j)
    {
        if (j < n_iter) // the same for every thread of the block
        {
            // p12, p22 of the row above the thread's first row
            float4 up12 = make_float4(0.f, 0.f, 0.f, 0.f), up22 = up12;
            if (tr > 0)
            {
                up12 = *(const float4*)&s_p12[srow - kRowsW];
                up22 = *(const float4*)&s_p22[srow - kRowsW];
            }
            float u12[4], u22[4];
            spread(up12, u12), spread(up22, u22);
#pragma unroll
            for (int i = 0; i < RPT; ++i)
            {
                const int gy = gy0 + i;
                const float l11 = lane_left(q11[i][3]), l21 = lane_left(q21[i][3]);
                float n1[4], n2[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    tvl1_u_values(cgx[i][k], cgy[i][k], cg[i][k], crc[i][k], a[i][k], b[i][k], q11[i][k], q12[i][k], q21[i][k],
                                  q22[i][k], k == 0 ? l11 : q11[i][k - 1], k == 0 ? l21 : q21[i][k - 1],
                                  i == 0 ? u12[k] : q12[i - 1][k], i == 0 ? u22[k] : q22[i - 1][k], gx0 + k > 0, gy > 0, l_t,
                                  theta, n1[k], n2[k]);
                const int cy = tr * RPT + i;
                if (inside[i] && tile_col && cy >= KT && cy < KT + TH)
                {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        e[j] += error_term(n1[k] - a[i][k], n2[k] - b[i][k]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    a[i][k] = n1[k], b[i][k] = n2[k];
            }
            *(float4*)&s_n1[srow] = make_float4(a[0][0], a[0][1], a[0][2], a[0][3]);
            *(float4*)&s_n2[srow] = make_float4(b[0][0], b[0][1], b[0][2], b[0][3]);
        }
        __syncthreads();
        if (j < n_iter)
        {
            // new u of the row below the thread's last row
            float4 dn1 = make_float4(0.f, 0.f, 0.f, 0.f), dn2 = dn1;
            if (tr + 1 < TR)
            {
                dn1 = *(const float4*)&s_n1[srow + kRowsW];
                dn2 = *(const float4*)&s_n2[srow + kRowsW];
            }
            float d1v[4], d2v[4];
            spread(dn1, d1v), spread(dn2, d2v);
#pragma unroll
            for (int i = 0; i < RPT; ++i)
            {
                const int gy = gy0 + i;
                const bool has_d = gy < rows - 1;
                const float r1e = lane_right(a[i][0]), r2e = lane_right(b[i][0]);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                {
                    const bool has_r = gx0 + k < cols - 1;
                    const float n1 = a[i][k], n2 = b[i][k];
                    const float r1 = k < 3 ? a[i][k < 3 ? k + 1 : 3] : r1e, r2 = k < 3 ? b[i][k < 3 ? k + 1 : 3] : r2e;
                    const float d1 = i + 1 < RPT ? a[i + 1 < RPT ? i + 1 : i][k] : d1v[k];
                    const float d2 = i + 1 < RPT ? b[i + 1 < RPT ? i + 1 : i][k] : d2v[k];
                    const float u1x = has_r ? r1 - n1 : 0.0f, u1y = has_d ? d1 - n1 : 0.0f;
                    const float u2x = has_r ? r2 - n2 : 0.0f, u2y = has_d ? d2 - n2 : 0.0f;
                    const float g1 = sqrtf(u1x * u1x + u1y * u1y), g2 = sqrtf(u2x * u2x + u2y * u2y);
                    const float ng1 = 1.0f + taut * g1, ng2 = 1.0f + taut * g2;
                    q11[i][k] = (q11[i][k] + taut * u1x) / ng1;
                    q12[i][k] = (q12[i][k] + taut * u1y) / ng1;
                    q21[i][k] = (q21[i][k] + taut * u2x) / ng2;
                    q22[i][k] = (q22[i][k] + taut * u2y) / ng2;
                }
            }
            *(float4*)&s_p12[srow] = make_float4(q12[RPT - 1][0], q12[RPT - 1][1], q12[RPT - 1][2], q12[RPT - 1][3]);
            *(float4*)&s_p22[srow] = make_float4(q22[RPT - 1][0], q22[RPT - 1][1], q22[RPT - 1][2], q22[RPT - 1][3]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i)
    {
        const int cy = tr * RPT + i;
        if (inside[i] && tile_col && cy >= KT && cy < KT + TH)
        {
            const size_t o = (size_t)(gy0 + i) * cols + gx0;
            *(float4*)(wu1 + o) = make_float4(a[i][0], a[i][1], a[i][2], a[i][3]);
            *(float4*)(wu2 + o) = make_float4(b[i][0], b[i][1], b[i][2], b[i][3]);
            *(float4*)(wp11 + o) = make_float4(q11[i][0], q11[i][1], q11[i][2], q11[i][3]);
            *(float4*)(wp12 + o) = make_float4(q12[i][0], q12[i][1], q12[i][2], q12[i][3]);
            *(float4*)(wp21 + o) = make_float4(q21[i][0], q21[i][1], q21[i][2], q21[i][3]);
            *(float4*)(wp22 + o) = make_float4(q22[i][0], q22[i][1], q22[i][2], q22[i][3]);
        }
    }
    if (redo)
        return;
#pragma unroll
    for (int j = 0; j < KT; ++j)
    {
        unsigned long long v = e[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            v += __shfl_down(v, o, 64);
        if ((tid & 63) == 0)
            s_err[j][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < KT && tid < n_iter)
    {
        unsigned long long sum = 0;
#pragma unroll
        for (int wv = 0; wv < NT / 64; ++wv)
            sum += s_err[tid][wv];
        atomicAdd(&err[it0 + tid], sum);
    }
}

// cv::resize(INTER_LINEAR) of the two flow planes to the next finer level, times 2: destination pixel (dx, dy)
__device__ __forceinline__ void upsample_at(const float* __restrict__ s1, const float* __restrict__ s2, int srows,
                                            int scols, float* __restrict__ d1, float* __restrict__ d2, int drows,
                                            int dcols, int dx, int dy)
{
    const double scx = 1.0 / ((double)dcols / scols), scy = 1.0 / ((double)drows / srows);
    float fx = (float)((dx + 0.5) * scx - 0.5);
    int sx = (int)floor((double)fx);
    fx -= sx;
    if (sx < 0)
        fx = 0, sx = 0;
    if (sx >= scols - 1)
        fx = 0, sx = scols - 1;
    float fy = (float)((dy + 0.5) * scy - 0.5);
    const int sy = (int)floor((double)fy);
    fy -= sy;
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const int y0 = sy < 0 ? 0 : (sy > srows - 1 ? srows - 1 : sy);
    const int y1 = sy + 1 < 0 ? 0 : (sy + 1 > srows - 1 ? srows - 1 : sy + 1);
    const bool last = sx >= scols - 1;
    const size_t o = (size_t)dy * dcols + dx;
    {
        const float* R0 = s1 + (size_t)y0 * scols;
        const float* R1 = s1 + (size_t)y1 * scols;
        const float h0 = last ? R0[scols - 1] : R0[sx] * a0 + R0[sx + 1] * a1;
        const float h1 = last ? R1[scols - 1] : R1[sx] * a0 + R1[sx + 1] * a1;
        d1[o] = (h0 * b0 + h1 * b1) * 2.0f;
    }
    {
        const float* R0 = s2 + (size_t)y0 * scols;
        const float* R1 = s2 + (size_t)y1 * scols;
        const float h0 = last ? R0[scols - 1] : R0[sx] * a0 + R0[sx + 1] * a1;
        const float h1 = last ? R1[scols - 1] : R1[sx] * a0 + R1[sx + 1] * a1;
        d2[o] = (h0 * b0 + h1 * b1) * 2.0f;
    }
}

// First iteration of a phase whose error is at most thr, looked for by one wave; 0x7FFFFFFF if there is none
// among the first `n` entries.
__device__ __forceinline__ int first_converged(const unsigned long long* __restrict__ err, int n, unsigned long long thr,
                                               int lane)
{
    int first = 0x7FFFFFFF;
    for (int j = lane; j < n; j += 64)
        if (err[j] <= thr)
        {
            first = j;
            break;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        first = min(first, __shfl_xor(first, o, 64));
    return first;
}

// What stands between two phases in one launch, one thread per pixel of the finished phase's level.
//   The gate: has the phase before `phase` finished within the n_enq launches of kt iterations queued for it?  It
//   has when an iteration met the threshold and the launch that holds it either ended with it or is followed by
//   another queued launch (which repairs the overrun), or when all `iters` iterations were queued and none met
//   it.  Otherwise *stop becomes `phase` (unless an earlier phase already stopped the chain) and nothing else
//   happens.  Every block evaluates the rule alike; block 0 records the stop.
//   The settling: a finished phase leaves its result in the buffer set its last launch wrote, set
//   (last_launch + 1) & 1.  The next phase reads set 0, so a result in set 1 is copied over (a pixel's two flow
//   values and four dual values).
//   By MODE, 1: the warp that opens the next phase of the same level (it reads the flow of its own pixel only,
//   so it can follow the settling without a grid-wide wait); 2: the pair's result, the interleaved flow.
// MODE 0 is gate and settling alone (the next phase is on another level).
struct Boundary
{
    const unsigned long long* err_prev;
    int n_enq, kt, iters;
    unsigned long long thr;
    int* stop;
    int phase;
    float *u1, *u2;
    const float* u_alt;
    float* p;
    const float* p_alt;
    int rows, cols;
};
struct WarpPlanes
{
    const float *I0, *I1, *I1x, *I1y;
    float *I1wx, *I1wy, *grad, *rho_c;
};
template <int MODE>
__global__ __launch_bounds__(256) void k_boundary(Boundary B, WarpPlanes W, float2* __restrict__ flow_uv)
{
    if (*B.stop <= B.phase)
        return;
    __shared__ int s_first;
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int queued = min(B.n_enq * B.kt, B.iters);
    if (tid < 64)
    {
        const int f = first_converged(B.err_prev, queued, B.thr, tid);
        if (tid == 0)
            s_first = f;
    }
    __syncthreads();
    const int first = s_first;
    bool complete;
    if (first == 0x7FFFFFFF)
        complete = queued >= B.iters;
    else
        complete = first % B.kt == B.kt - 1 || first == B.iters - 1 || first / B.kt + 1 < B.n_enq;
    if (!complete)
    {
        if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0)
            *B.stop = B.phase;
        return;
    }
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= B.cols || y >= B.rows)
        return;
    const size_t n = (size_t)B.rows * B.cols, i = (size_t)y * B.cols + x;
    const int done = first == 0x7FFFFFFF ? B.iters : first + 1;
    const bool in_alt = (((done - 1) / B.kt + 1) & 1) != 0;
    float a, b;
    if (in_alt)
    {
        a = B.u_alt[i], b = B.u_alt[n + i];
        B.u1[i] = a, B.u2[i] = b;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            B.p[j * n + i] = B.p_alt[j * n + i];
    }
    else if (MODE != 0)
        a = B.u1[i], b = B.u2[i];
    if (MODE == 1)
        warp_at(W.I0, W.I1, W.I1x, W.I1y, a, b, x, y, B.rows, B.cols, W.I1wx, W.I1wy, W.grad, W.rho_c);
    if (MODE == 2)
        flow_uv[i] = make_float2(a, b);
}

// What opens a level, one thread per pixel: the centred gradient of I1, p = 0, and the starting flow - the flow of
// the level before brought up to this one (upsample_at), or zero on the coarsest level (s1 null).
__global__ void k_open_level(const float* __restrict__ I1, int rows, int cols, float* __restrict__ dx,
                             float* __restrict__ dy, float* __restrict__ p, const float* __restrict__ s1,
                             const float* __restrict__ s2, int srows, int scols, float* __restrict__ d1,
                             float* __restrict__ d2, Guard guard)
{
    if (stopped(guard))
        return;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows)
        return;
    const size_t n = (size_t)rows * cols, i = (size_t)y * cols + x;
    {
        const int xm = x > 0 ? x - 1 : 0, xp = x < cols - 1 ? x + 1 : cols - 1;
        const int ym = y > 0 ? y - 1 : 0, yp = y < rows - 1 ? y + 1 : rows - 1;
        dx[i] = 0.5f * (I1[(size_t)y * cols + xp] - I1[(size_t)y * cols + xm]);
        dy[i] = 0.5f * (I1[(size_t)yp * cols + x] - I1[(size_t)ym * cols + x]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        p[j * n + i] = 0.0f;
    if (!s1)
    {
        d1[i] = 0.0f, d2[i] = 0.0f;
        return;
    }
    upsample_at(s1, s2, srows, scols, d1, d2, rows, cols, x, y);
}

inline dim3 grid2(int cols, int rows, dim3 b)
{
    return dim3((cols + b.x - 1) / b.x, (rows + b.y - 1) / b.y);
}

#define SRX_LAUNCH_CHECK(what)                                                                                         \
    do                                                                                                                 \
    {                                                                                                                  \
        const hipError_t e__ = hipGetLastError();                                                                      \
        if (e__ != hipSuccess)                                                                                         \
            return hip_fail(e__, what);                                                                                \
    } while (0)

// k_tvl1_rows: iterations per launch and threads per block for levels above / up to a megapixel.  Tuning aids:
// SRX_TVL1_ROWS="kt_big,nt_big,kt_mid,nt_mid" in the environment replaces them (read once).
int g_rows_kt_big = 2, g_rows_nt_big = 512, g_rows_kt_mid = 4, g_rows_nt_mid = 512;
// launches queued for a phase beyond what the previous pair needed: max(min, need / div); SRX_TVL1_MARGIN="min,div"
int g_chain_margin_min = 3, g_chain_margin_div = 3;
const bool g_rows_env = [] {
    if (const char* e = std::getenv("SRX_TVL1_ROWS"))
    {
        int v[4];
        if (std::sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4)
            g_rows_kt_big = v[0], g_rows_nt_big = v[1], g_rows_kt_mid = v[2], g_rows_nt_mid = v[3];
    }
    if (const char* e = std::getenv("SRX_TVL1_MARGIN"))
    {
        int a = 0, b = 0;
        if (std::sscanf(e, "%d,%d", &a, &b) == 2 && a >= 0 && b >= 1)
            g_chain_margin_min = a, g_chain_margin_div = b;
    }
    return true;
}();

constexpr int kMaxScales = 16;
constexpr int kPollStride = 32; // most inner iterations enqueued before the first look at the error
constexpr int kPollRest = 12;   // inner iterations enqueued between two later looks

} // namespace

// frame (device, dense) -> 8-bit grey as float; depth: SRX_8U / SRX_16U / SRX_32F
int launch_to_gray(const void* raw, size_t px, int ch, int depth, float* gray, hipStream_t st)
{
    hipLaunchKernelGGL(k_to_gray, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, st, raw, px, ch,
                       depth == SRX_32F ? 1 : (depth == SRX_16U ? 2 : 0), gray);
    SRX_LAUNCH_CHECK("to_gray");
    return SRX_OK;
}

struct Tvl1::Impl
{
    struct Level
    {
        int rows = 0, cols = 0;
        DeviceBuffer I0, I1, u1, u2;
    };
    Level lv[kMaxScales];
    DeviceBuffer I1x, I1y, I1wx, I1wy, grad, rho_c, p, err, raw0, raw1;
    DeviceBuffer u_alt, p_alt; // second set of u (2 planes) and p (4 planes) for the ping-pong of k_tvl1_iter
    PinnedBuffer host_err;
    // inner iterations each (level, warp) ran in the previous call: consecutive frame pairs of a stream converge
    // alike, so this sizes the first batch of launches (results do not depend on it, only the number of
    // launches that find the iteration already converged)
    std::vector<int> last_done, before_last_done;
    long long pixel_iterations = 0; // sum over (level, warp) of pixels x executed inner iterations, last call
};

int Tvl1::calc_images(const srx_image& f0, const srx_image& f1, const Tvl1Params& p, float* flow_uv,
                      long long* executed_iters, hipStream_t st)
{
    if (!impl_)
        impl_ = std::make_shared<Impl>();
    Impl& m = *impl_;
    const int rows = f0.rows, cols = f0.cols, ch = f0.channels;
    if (ch != 1 && ch != 3 && ch != 4)
    {
        set_error("flow input must have 1, 3 or 4 channels");
        return SRX_ERR_INVALID;
    }
    const size_t px = (size_t)rows * cols;
    const size_t esz = f0.depth == SRX_8U ? 1 : (f0.depth == SRX_16U ? 2 : 4);
    const size_t rowbytes = (size_t)cols * ch * esz;
    if (int rc = m.lv[0].I0.reserve(px * sizeof(float)))
        return rc;
    if (int rc = m.lv[0].I1.reserve(px * sizeof(float)))
        return rc;
    const srx_image* fr[2] = {&f0, &f1};
    DeviceBuffer* raw[2] = {&m.raw0, &m.raw1};
    float* gray[2] = {m.lv[0].I0.as<float>(), m.lv[0].I1.as<float>()};
    for (int i = 0; i < 2; ++i)
    {
        if (int rc = raw[i]->reserve(rowbytes * rows))
            return rc;
        const size_t step = fr[i]->step ? fr[i]->step : rowbytes;
        SRX_HIP(copy_pitched(raw[i]->get(), rowbytes, fr[i]->data, step, rowbytes, rows,
                             fr[i]->mem == SRX_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_to_gray, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, st, raw[i]->get(), px, ch,
                           f0.depth == SRX_32F ? 1 : (f0.depth == SRX_16U ? 2 : 0), gray[i]);
        SRX_LAUNCH_CHECK("to_gray");
    }
    return calc_gray(nullptr, nullptr, rows, cols, p, flow_uv, executed_iters, st);
}

// I0 / I1 == nullptr: the level-0 images are already in place (calc_images)
int Tvl1::calc_gray(const float* I0, const float* I1, int rows, int cols, const Tvl1Params& p, float* flow_uv,
                    long long* executed_iters, hipStream_t st)
{
    if (!impl_)
        impl_ = std::make_shared<Impl>();
    Impl& m = *impl_;
    if (rows < 1 || cols < 1 || !flow_uv)
    {
        set_error("tvl1: bad arguments");
        return SRX_ERR_INVALID;
    }
    const dim3 b2(64, 4);
    int nscales = std::min(std::max(p.nscales, 1), kMaxScales);
    m.lv[0].rows = rows, m.lv[0].cols = cols;
    const size_t px0 = (size_t)rows * cols;
    if (I0)
    {
        if (int rc = m.lv[0].I0.reserve(px0 * sizeof(float)))
            return rc;
        if (int rc = m.lv[0].I1.reserve(px0 * sizeof(float)))
            return rc;
        SRX_HIP(hipMemcpyAsync(m.lv[0].I0.get(), I0, px0 * sizeof(float), hipMemcpyDeviceToDevice, st));
        SRX_HIP(hipMemcpyAsync(m.lv[0].I1.get(), I1, px0 * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    for (int s = 1; s < nscales; ++s)
    {
        const int r = (m.lv[s - 1].rows + 1) / 2, c = (m.lv[s - 1].cols + 1) / 2;
        if (c < 16 || r < 16)
        {
            nscales = s;
            break;
        }
        Impl::Level& L = m.lv[s];
        L.rows = r, L.cols = c;
        if (int rc = L.I0.reserve((size_t)r * c * sizeof(float)))
            return rc;
        if (int rc = L.I1.reserve((size_t)r * c * sizeof(float)))
            return rc;
        dim3 g2 = grid2(c, r, b2);
        g2.z = 2;
        hipLaunchKernelGGL(k_pyr_down, g2, b2, 0, st, m.lv[s - 1].I0.as<float>(), m.lv[s - 1].I1.as<float>(),
                           m.lv[s - 1].rows, m.lv[s - 1].cols, L.I0.as<float>(), L.I1.as<float>(), r, c);
        SRX_LAUNCH_CHECK("pyr_down");
    }
    for (int s = 0; s < nscales; ++s)
    {
        const size_t n = (size_t)m.lv[s].rows * m.lv[s].cols * sizeof(float);
        if (int rc = m.lv[s].u1.reserve(n))
            return rc;
        if (int rc = m.lv[s].u2.reserve(n))
            return rc;
    }
    for (DeviceBuffer* d : {&m.I1x, &m.I1y, &m.I1wx, &m.I1wy, &m.grad, &m.rho_c})
        if (int rc = d->reserve(px0 * sizeof(float)))
            return rc;
    if (int rc = m.p.reserve(4 * px0 * sizeof(float)))
        return rc;
    if (int rc = m.p_alt.reserve(4 * px0 * sizeof(float)))
        return rc;
    if (int rc = m.u_alt.reserve(2 * px0 * sizeof(float)))
        return rc;
    const int iters = std::max(p.iterations, 1);
    const int n_phases = nscales * p.warps;
    // one error array per phase, and behind them the word that stops the chain (see Guard)
    const size_t err_words = (size_t)n_phases * iters;
    if (int rc = m.err.reserve((err_words + 1) * sizeof(unsigned long long)))
        return rc;
    if (int rc = m.host_err.reserve((err_words + 1) * sizeof(unsigned long long)))
        return rc;
    unsigned long long* const derr = m.err.as<unsigned long long>();
    int* const dstop = (int*)(derr + err_words);
    const unsigned long long* const herr = (const unsigned long long*)m.host_err.get();
    const int* const hstop = (const int*)(herr + err_words);

    if (m.last_done.size() != (size_t)n_phases)
    {
        m.last_done.assign((size_t)n_phases, kPollStride - 1);
        m.before_last_done.assign((size_t)n_phases, 0);
    }
    const float l_t = (float)(p.lambda * p.theta), taut = (float)(p.tau / p.theta), theta = (float)p.theta;

    // ---- the plan: one phase per (level, warp), coarsest level first
    struct Phase
    {
        int s, w, R, C, kernel, kt, rows_nt, n_launch, n_enq;
        size_t n;
        unsigned long long thr;
        unsigned long long* err;
    };
    std::vector<Phase> plan((size_t)n_phases);
    for (int s = nscales - 1, k = 0; s >= 0; --s)
        for (int w = 0; w < p.warps; ++w, ++k)
        {
            Phase& ph = plan[(size_t)k];
            ph.s = s, ph.w = w, ph.R = m.lv[s].rows, ph.C = m.lv[s].cols;
            ph.n = (size_t)ph.R * ph.C;
            const float scaled_eps = (float)(p.epsilon * p.epsilon * (double)ph.n);
            ph.thr = (unsigned long long)((double)scaled_eps * 1073741824.0);
            ph.err = derr + (size_t)k * iters;
            // 0: at most 200 000 pixels, 1: at most 1 000 000, 2: more
            const int by_size = ph.n > (size_t)1000000 ? 2 : (ph.n > (size_t)200000 ? 1 : 0);
            // Iterations per launch.  Small levels are bound by the time a launch takes whatever it does: 8 iterations
            // on tiles of 16 x 16.  Larger levels are bound by memory traffic: where the width is a multiple of 4 they
            // run several iterations per launch on regions of whole 512-byte rows (k_tvl1_rows), which reads and
            // writes every plane once per launch; other widths fall back to 32 x 32 tiles (up to a megapixel) and to
            // one iteration per launch (above: the short row segments of a tile with halo waste too much of every
            // cache line there).
            const bool wide = ph.C % 4 == 0;
            // kernel: 0 k_tvl1_multi<8,16,16>, 1 k_tvl1_multi<4,32,32>, 2 one iteration per launch, 3 k_tvl1_rows
            ph.kernel = p.iterationsPerLaunch == 1 ? 2 : p.iterationsPerLaunch == 4 ? 1 : p.iterationsPerLaunch == 8 ? 0
                        : (p.iterationsPerLaunch == 12 || p.iterationsPerLaunch == 14) ? (wide ? 3 : 1)
                        : (by_size >= 1 && wide ? 3 : by_size);
            const int rows_kt = p.iterationsPerLaunch == 12 ? 2 : p.iterationsPerLaunch == 14 ? 4
                                : (by_size == 2 ? g_rows_kt_big : g_rows_kt_mid);
            ph.rows_nt = by_size == 2 ? g_rows_nt_big : g_rows_nt_mid;
            ph.kt = ph.kernel == 3 ? rows_kt : ph.kernel == 2 ? 1 : (ph.kernel == 1 ? 4 : 8);
            ph.n_launch = (iters + ph.kt - 1) / ph.kt;
            ph.n_enq = 0;
        }

    auto launch_iterations = [&](const Phase& ph, int k, int bidx) {
        Impl::Level& L = m.lv[ph.s];
        const size_t n = ph.n;
        const int R = ph.R, C = ph.C, kt = ph.kt;
        float* ua[2][2] = {{L.u1.as<float>(), L.u2.as<float>()}, {m.u_alt.as<float>(), m.u_alt.as<float>() + n}};
        float* pa[2] = {m.p.as<float>(), m.p_alt.as<float>()};
        const int src = bidx & 1, dst = src ^ 1;
        Tvl1Planes P;
        P.I1wx = m.I1wx.as<float>(), P.I1wy = m.I1wy.as<float>(), P.grad = m.grad.as<float>();
        P.rho_c = m.rho_c.as<float>();
        P.u1 = ua[src][0], P.u2 = ua[src][1];
        P.p11 = pa[src], P.p12 = pa[src] + n, P.p21 = pa[src] + 2 * n, P.p22 = pa[src] + 3 * n;
        P.nu1 = ua[dst][0], P.nu2 = ua[dst][1];
        P.np11 = pa[dst], P.np12 = pa[dst] + n, P.np21 = pa[dst] + 2 * n, P.np22 = pa[dst] + 3 * n;
        const Guard g{dstop, k};
        // (a launch past the last iteration only ever repairs the one before it)
        const int first = bidx * kt, count = std::max(0, std::min(kt, iters - first));
        const int tw = ph.kernel == 0 ? 16 : 32, th = tw;
        const dim3 gm((C + tw - 1) / tw, (R + th - 1) / th);
        if (ph.kernel == 2) // one iteration per launch: launch `bidx` is iteration `bidx`
        {
            if (C % 4 == 0)
                hipLaunchKernelGGL(k_tvl1_iter4, grid2(C, R, dim3(kIt4X, kIt4Y)), dim3(32, kIt4Y + 2), 0, st, P, R, C, l_t,
                                   theta, taut, ph.err, bidx, ph.thr, g);
            else
                hipLaunchKernelGGL(k_tvl1_iter, grid2(C, R, dim3(kItX, kItY)), dim3(kItX, kItY), 0, st, P, R, C, l_t, theta,
                                   taut, ph.err, bidx, ph.thr, g);
        }
        else if (ph.kernel == 3)
        {
#define SRX_ROWS(KTV, RPTV, NTV)                                                                                       \
    hipLaunchKernelGGL((k_tvl1_rows<KTV, RPTV, NTV>),                                                                  \
                       dim3((C + kRowsTW - 1) / kRowsTW, (R + (NTV / 32 * RPTV - 2 * KTV) - 1) / (NTV / 32 * RPTV - 2 * KTV)), \
                       dim3(NTV), 0, st, P, R, C, l_t, theta, taut, ph.err, iters, first, count, ph.thr, g)
            if (kt == 2 && ph.rows_nt == 256)
                SRX_ROWS(2, 2, 256);
            else if (kt == 2 && ph.rows_nt == 512)
                SRX_ROWS(2, 2, 512);
            else if (kt == 2)
                SRX_ROWS(2, 2, 1024);
            else if (ph.rows_nt == 256)
                SRX_ROWS(4, 2, 256);
            else if (ph.rows_nt == 512)
                SRX_ROWS(4, 2, 512);
            else
                SRX_ROWS(4, 2, 1024);
#undef SRX_ROWS
        }
        else if (ph.kernel == 1)
            hipLaunchKernelGGL((k_tvl1_multi<4, 32, 32, 256>), gm, dim3(256), 0, st, P, R, C, l_t, theta, taut, ph.err, iters,
                               first, count, ph.thr, g);
        else
            hipLaunchKernelGGL((k_tvl1_multi<8, 16, 16, 1024>), gm, dim3(1024), 0, st, P, R, C, l_t, theta, taut, ph.err,
                               iters, first, count, ph.thr, g);
    };
    // What stands between phase k - 1 and phase k and opens phase k.  Within a level: one launch (gate, settling,
    // warp).  Between levels: gate and settling, then the new level's gradient, p = 0 and starting flow, then the
    // warp (it gathers the gradient around its source, so it waits for the whole plane).  Behind the last phase:
    // gate, settling and the interleaved result.
    auto launch_step = [&](int k) {
        WarpPlanes W{};
        if (k < n_phases)
        {
            Impl::Level& N = m.lv[plan[(size_t)k].s];
            W = WarpPlanes{N.I0.as<float>(), N.I1.as<float>(), m.I1x.as<float>(), m.I1y.as<float>(),
                           m.I1wx.as<float>(), m.I1wy.as<float>(), m.grad.as<float>(), m.rho_c.as<float>()};
        }
        const bool same_level = k > 0 && k < n_phases && plan[(size_t)k].w > 0;
        if (k > 0)
        {
            const Phase& prev = plan[(size_t)k - 1];
            Impl::Level& L = m.lv[prev.s];
            const Boundary B{prev.err, prev.n_enq, prev.kt, iters, prev.thr, dstop, k, L.u1.as<float>(), L.u2.as<float>(),
                             m.u_alt.as<float>(), m.p.as<float>(), m.p_alt.as<float>(), prev.R, prev.C};
            const dim3 g2 = grid2(prev.C, prev.R, b2);
            if (k == n_phases)
                hipLaunchKernelGGL(k_boundary<2>, g2, b2, 0, st, B, W, (float2*)flow_uv);
            else if (same_level)
                hipLaunchKernelGGL(k_boundary<1>, g2, b2, 0, st, B, W, (float2*)nullptr);
            else
                hipLaunchKernelGGL(k_boundary<0>, g2, b2, 0, st, B, W, (float2*)nullptr);
        }
        if (k == n_phases || same_level)
            return;
        const Phase& ph = plan[(size_t)k];
        Impl::Level& N = m.lv[ph.s];
        const Guard g{dstop, k};
        const dim3 g2 = grid2(ph.C, ph.R, b2);
        const Impl::Level* below = k > 0 ? &m.lv[plan[(size_t)k - 1].s] : nullptr;
        hipLaunchKernelGGL(k_open_level, g2, b2, 0, st, N.I1.as<float>(), ph.R, ph.C, m.I1x.as<float>(), m.I1y.as<float>(),
                           m.p.as<float>(), below ? below->u1.as<float>() : (const float*)nullptr,
                           below ? below->u2.as<float>() : (const float*)nullptr, below ? below->rows : 0,
                           below ? below->cols : 0, N.u1.as<float>(), N.u2.as<float>(), g);
        hipLaunchKernelGGL(k_warp, g2, b2, 0, st, W.I0, W.I1, W.I1x, W.I1y, N.u1.as<float>(), N.u2.as<float>(), ph.R, ph.C,
                           W.I1wx, W.I1wy, W.grad, W.rho_c, g);
    };

    SRX_HIP(hipMemsetAsync(derr, 0, (err_words + 1) * sizeof(unsigned long long), st));
    // Launches of a phase queued at first: what the same phase of the previous pair needed (consecutive pairs of a
    // stream converge alike) and a margin; launches queued beyond the converged iteration return at once.  The host
    // looks once, behind the whole chain; a chain that stopped at some phase is continued from there.
    // launch n_launch of a kernel of several iterations only ever repairs an overrun of the launch before it; a
    // kernel of one iteration per launch has no such launch (and would write behind its error array)
    auto launch_cap = [](const Phase& ph) { return ph.n_launch + (ph.kt > 1 ? 1 : 0); };
    auto first_batch = [&](const Phase& ph, int predicted) {
        const int need = (std::max(predicted, 1) + ph.kt - 1) / ph.kt;
        return std::min(launch_cap(ph), need + std::max(g_chain_margin_min, need / g_chain_margin_div));
    };
    int resume = 0;         // first phase whose opening has not been queued yet
    int unfinished = -1;    // phase that has to go on before `resume` (its opening is done), or -1
    for (int round = 0;; ++round)
    {
        static const int kRun = 0x7FFFFFFF;
        SRX_HIP(hipMemcpyAsync(dstop, &kRun, sizeof(int), hipMemcpyHostToDevice, st));
        if (unfinished >= 0)
        {
            Phase& ph = plan[(size_t)unfinished];
            const int more = std::min(launch_cap(ph) - ph.n_enq, std::max(ph.n_enq, std::max(kPollRest / ph.kt, 2)));
            for (int b = ph.n_enq; b < ph.n_enq + more; ++b)
                launch_iterations(ph, unfinished, b);
            ph.n_enq += more;
        }
        for (int k = resume; k <= n_phases; ++k)
        {
            launch_step(k);
            if (k == n_phases)
                break;
            Phase& ph = plan[(size_t)k];
            ph.n_enq = first_batch(ph, std::max(m.last_done[(size_t)k], m.before_last_done[(size_t)k]));
            for (int b = 0; b < ph.n_enq; ++b)
                launch_iterations(ph, k, b);
        }
        SRX_LAUNCH_CHECK("tvl1 chain");
        SRX_HIP(hipMemcpyAsync(m.host_err.get(), derr, (err_words + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        SRX_HIP(hipStreamSynchronize(st));
        const int stop = *hstop;
        if (stop == kRun)
            break;
        if (stop < 1 || stop > n_phases || round > 4 * n_phases + 64)
        {
            set_error("tvl1: the launch chain stopped at an impossible phase");
            return SRX_ERR_DEVICE;
        }
        // phase stop - 1 needs more launches; everything from `stop` on (its gate included) is queued again
        unfinished = stop - 1;
        resume = stop;
    }
    long long executed = 0;
    m.pixel_iterations = 0;
    for (int k = 0; k < n_phases; ++k)
    {
        const Phase& ph = plan[(size_t)k];
        int done = iters;
        for (int it = 0; it < iters; ++it)
            if (herr[(size_t)k * iters + it] <= ph.thr)
            {
                done = it + 1;
                break;
            }
        executed += done;
        m.pixel_iterations += (long long)done * (long long)ph.n;
        m.before_last_done[(size_t)k] = m.last_done[(size_t)k];
        m.last_done[(size_t)k] = done;
    }
    if (executed_iters)
        *executed_iters = executed;
    return SRX_OK;
}

long long Tvl1::pixel_iterations() const
{
    return impl_ ? impl_->pixel_iterations : 0;
}

void Tvl1::release()
{
    impl_.reset();
}

} // namespace srx
