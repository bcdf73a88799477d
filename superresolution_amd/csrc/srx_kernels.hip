// HIP kernels of the BTV-L1 solver for gfx950 (MI355X).
//
// Build with -ffp-contract=off: the solver's sign() steps make the iteration
// sensitive to the last bit, so every float operation below is meant to round
// exactly as written (same order as the OpenCV routines the reference calls).
//
// Data layout in HBM
//   images      float32, channels interleaved, dense rows
//   index maps  int16 (x, y) per HR pixel: the source index cv::remap(INTER_NEAREST)
//               derives from the float map (round half to even, saturate)
//   signs       one int8x4 per LR pixel and window frame (bytes 0..ch-1 used)
#include "srx_common.h"

namespace srx
{

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

__device__ __forceinline__ int clampi(int v, int lo, int hi)
{
    return v < lo ? lo : (v > hi ? hi : v);
}

__device__ __forceinline__ float dsign(float a, float b)
{
    return a > b ? 1.0f : (a < b ? -1.0f : 0.0f);
}

// saturate_cast<short>(cvRound(v)): round half to even; NaN and values beyond
// the int range give SHRT_MIN (cvtsd2si's "integer indefinite", then saturated)
__device__ __forceinline__ short round_sat16(float v)
{
    if (!(fabsf(v) < 2147483648.0f))
        return (short)-32768;
    float r = rintf(v);
    r = fminf(fmaxf(r, -32768.0f), 32767.0f);
    return (short)(int)r;
}

// ---------------------------------------------------------------------------
// calcRelativeMotions (lib/src/btv_l1.cpp:36-57): running sums away from base
__global__ void k_relative_motions(PtrList fwd, MutPtrList rel, int n, int base_idx, size_t count)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count)
        return;
    ((float*)rel.p[base_idx])[e] = 0.0f;
    float acc = 0.0f;
    for (int i = base_idx - 1; i >= 0; --i)
    {
        acc = acc + ((const float*)fwd.p[i])[e];
        ((float*)rel.p[i])[e] = acc;
    }
    acc = 0.0f;
    for (int i = base_idx + 1; i < n; ++i)
    {
        acc = acc - ((const float*)fwd.p[i - 1])[e];
        ((float*)rel.p[i])[e] = acc;
    }
}

// ---------------------------------------------------------------------------
// One bicubic sample in cv::resize order: horizontal taps on 4 source rows
// (float intermediate), then the vertical taps.
template <int CH>
__device__ __forceinline__ void cubic_sample(const float* __restrict__ src, int srows, int scols, int sx, float4 a,
                                             int sy, float4 b, float (&out)[CH])
{
    const int x0 = clampi(sx - 1, 0, scols - 1), x1 = clampi(sx, 0, scols - 1);
    const int x2 = clampi(sx + 1, 0, scols - 1), x3 = clampi(sx + 2, 0, scols - 1);
    float h[4][CH];
#pragma unroll
    for (int r = 0; r < 4; ++r)
    {
        const float* S = src + (size_t)clampi(sy - 1 + r, 0, srows - 1) * scols * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            h[r][c] = S[x0 * CH + c] * a.x + S[x1 * CH + c] * a.y + S[x2 * CH + c] * a.z + S[x3 * CH + c] * a.w;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
        out[c] = h[0][c] * b.x + h[1][c] * b.y + h[2][c] * b.z + h[3][c] * b.w;
}

// upscaleMotions + buildMotionMaps + remap's float->short index conversion
// (lib/src/btv_l1.cpp:59-101; cv::remap INTER_NEAREST)
__global__ void k_build_maps(const float* __restrict__ rel, int rows, int cols, int scale, CubicTables t,
                             short2* __restrict__ fwd_map, short2* __restrict__ bwd_map)
{
    const int R = rows * scale, C = cols * scale;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= C || y >= R)
        return;
    float m[2];
    cubic_sample<2>(rel, rows, cols, t.xofs[x], t.alpha[x], t.yofs[y], t.beta[y], m);
    const float s = (float)scale;
    const float mx = m[0] * s, my = m[1] * s;
    const float fx = (float)x - mx, fy = (float)y - my;
    const float bx = (float)x + mx, by = (float)y + my;
    const size_t o = (size_t)y * C + x;
    fwd_map[o] = make_short2(round_sat16(fx), round_sat16(fy));
    bwd_map[o] = make_short2(round_sat16(bx), round_sat16(by));
}

template <int CH>
__global__ void k_resize_cubic(const float* __restrict__ src, int srows, int scols, CubicTables t,
                               float* __restrict__ dst, int drows, int dcols)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dcols || y >= drows)
        return;
    float v[CH];
    cubic_sample<CH>(src, srows, scols, t.xofs[x], t.alpha[x], t.yofs[y], t.beta[y], v);
    float* D = dst + ((size_t)y * dcols + x) * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c)
        D[c] = v[c];
}

// ---------------------------------------------------------------------------
// BTV regulariser at one pixel (lib/src/btv_l1.cpp:212-235), straight from
// global memory.  Border band of width `radius` stays zero.
template <int CH>
__device__ __forceinline__ void btv_at(const float* __restrict__ X, int H, int W, int i, int j, const BtvW& w,
                                       float (&reg)[CH])
{
#pragma unroll
    for (int c = 0; c < CH; ++c)
        reg[c] = 0.0f;
    const int p = w.radius;
    if (i < p || i >= H - p || j < p || j >= W - p)
        return;
    float v[CH];
    const float* P = X + ((size_t)i * W + j) * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c)
        v[c] = P[c];
    int ind = 0;
    for (int m = 0; m <= p; ++m)
        for (int l = p; l + m >= 0; --l, ++ind)
        {
            const float* P3 = X + ((size_t)(i + m) * W + (j + l)) * CH;
            const float* P2 = X + ((size_t)(i - m) * W + (j - l)) * CH;
            const float wt = w.w[ind];
#pragma unroll
            for (int c = 0; c < CH; ++c)
                reg[c] += wt * (dsign(v[c], P3[c]) - dsign(P2[c], v[c]));
        }
}

template <int CH>
__global__ void k_btv_regularization(const float* __restrict__ X, int H, int W, BtvW w, float* __restrict__ dst)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (j >= W || i >= H)
        return;
    float reg[CH];
    btv_at<CH>(X, H, W, i, j, w, reg);
    float* D = dst + ((size_t)i * W + j) * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c)
        D[c] = reg[c];
}

// ---------------------------------------------------------------------------
// Forward chain, generic in scale and blur size.
// a(Y, X) = Xhr(bwd(Y, X)) or 0 outside   (remap NEAREST, constant border)
template <int CH>
__device__ __forceinline__ void fetch_warped(const float* __restrict__ X, int H, int W,
                                             const short2* __restrict__ map, int Yr, int Xr, float (&o)[CH])
{
    const short2 m = map[(size_t)Yr * W + Xr];
    if ((unsigned)(int)m.x < (unsigned)W && (unsigned)(int)m.y < (unsigned)H)
    {
        const float* P = X + ((size_t)m.y * W + m.x) * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            o[c] = P[c];
    }
    else
    {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            o[c] = 0.0f;
    }
}

// Row pass of the separable Gaussian at (Yr, Xc) in OpenCV's operation order:
// n <= 5: centre*k0 + (l1+r1)*k1 + (l2+r2)*k2; n >= 7: taps left to right.
template <int CH>
__device__ __forceinline__ void warped_blur_row(const float* __restrict__ X, int H, int W,
                                                const short2* __restrict__ map, int Yr, int Xc, const GaussK& g,
                                                float (&r)[CH])
{
    const int n = g.n, hh = n >> 1;
    float t[CH], u[CH];
    if (n <= 5)
    {
        fetch_warped<CH>(X, H, W, map, Yr, Xc, t);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            r[c] = t[c] * g.k[hh];
        for (int j = 1; j <= hh; ++j)
        {
            fetch_warped<CH>(X, H, W, map, Yr, reflect101(Xc - j, W), t);
            fetch_warped<CH>(X, H, W, map, Yr, reflect101(Xc + j, W), u);
#pragma unroll
            for (int c = 0; c < CH; ++c)
                r[c] += (t[c] + u[c]) * g.k[hh + j];
        }
    }
    else
    {
        fetch_warped<CH>(X, H, W, map, Yr, reflect101(Xc - hh, W), t);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            r[c] = g.k[0] * t[c];
        for (int j = 1; j < n; ++j)
        {
            fetch_warped<CH>(X, H, W, map, Yr, reflect101(Xc - hh + j, W), t);
#pragma unroll
            for (int c = 0; c < CH; ++c)
                r[c] += g.k[j] * t[c];
        }
    }
}

// sign(y_k - D H M_k X) for every LR pixel and window frame (blockIdx.z = k)
// lib/src/btv_l1.cpp:428-434
template <int CH>
__global__ void k_forward_sign(const float* __restrict__ X, int H, int W, PtrList bwd_maps, PtrList ys,
                               MutPtrList signs, int scale, GaussK g)
{
    const int h = H / scale, w = W / scale;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = blockIdx.z;
    if (x >= w || y >= h)
        return;
    const short2* map = (const short2*)bwd_maps.p[k];
    const int Yc = y * scale, Xc = x * scale;
    const int hh = g.n >> 1;
    float acc[CH], ra[CH], rb[CH];
    // column pass: centre row first, then symmetric pairs outwards
    warped_blur_row<CH>(X, H, W, map, Yc, Xc, g, ra);
#pragma unroll
    for (int c = 0; c < CH; ++c)
        acc[c] = ra[c] * g.k[hh];
    for (int j = 1; j <= hh; ++j)
    {
        warped_blur_row<CH>(X, H, W, map, reflect101(Yc - j, H), Xc, g, ra);
        warped_blur_row<CH>(X, H, W, map, reflect101(Yc + j, H), Xc, g, rb);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            acc[c] += (ra[c] + rb[c]) * g.k[hh + j];
    }
    const float* yk = (const float*)ys.p[k] + ((size_t)y * w + x) * CH;
    char4 s = make_char4(0, 0, 0, 0);
    signed char sv[4] = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < CH; ++c)
        sv[c] = (signed char)dsign(yk[c], acc[c]);
    s.x = sv[0];
    s.y = sv[1];
    s.z = sv[2];
    s.w = sv[3];
    ((char4*)signs.p[k])[(size_t)y * w + x] = s;
}

// ---------------------------------------------------------------------------
// Transposed chain, generic.  B(fy, fx) = gauss(zero_insert(sign))(fy, fx)
// evaluated lazily: the zero-inserted image is non-zero only where both
// coordinates are multiples of scale.
template <int CH>
__device__ __forceinline__ void sign_at(const char4* __restrict__ sg, int w, int scale, int Yr, int Xr, float (&o)[CH])
{
    // caller guarantees Yr % scale == 0
    if (Xr % scale == 0)
    {
        const char4 s = sg[(size_t)(Yr / scale) * w + (Xr / scale)];
        const float sv[4] = {(float)s.x, (float)s.y, (float)s.z, (float)s.w};
#pragma unroll
        for (int c = 0; c < CH; ++c)
            o[c] = sv[c];
    }
    else
    {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            o[c] = 0.0f;
    }
}

template <int CH>
__device__ __forceinline__ void upsampled_blur_row(const char4* __restrict__ sg, int W, int scale, int Yr, int fx,
                                                   const GaussK& g, float (&r)[CH])
{
    if (Yr % scale != 0)
    {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            r[c] = 0.0f;
        return;
    }
    const int w = W / scale;
    const int n = g.n, hh = n >> 1;
    float t[CH], u[CH];
    if (n <= 5)
    {
        sign_at<CH>(sg, w, scale, Yr, fx, t);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            r[c] = t[c] * g.k[hh];
        for (int j = 1; j <= hh; ++j)
        {
            sign_at<CH>(sg, w, scale, Yr, reflect101(fx - j, W), t);
            sign_at<CH>(sg, w, scale, Yr, reflect101(fx + j, W), u);
#pragma unroll
            for (int c = 0; c < CH; ++c)
                r[c] += (t[c] + u[c]) * g.k[hh + j];
        }
    }
    else
    {
        sign_at<CH>(sg, w, scale, Yr, reflect101(fx - hh, W), t);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            r[c] = g.k[0] * t[c];
        for (int j = 1; j < n; ++j)
        {
            sign_at<CH>(sg, w, scale, Yr, reflect101(fx - hh + j, W), t);
#pragma unroll
            for (int c = 0; c < CH; ++c)
                r[c] += g.k[j] * t[c];
        }
    }
}

// X_out = X + tau * (sum_k M_k^T H^T D^T sign_k - lambda * btv(X))
// lib/src/btv_l1.cpp:437-452; both addWeighted steps evaluate in double.
template <int CH>
__global__ void k_backproject_update(const float* __restrict__ X, float* __restrict__ X_out, int H, int W,
                                     PtrList fwd_maps, PtrList signs, int K, int scale, GaussK g, BtvW bw, double tau,
                                     double lambda)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (j >= W || i >= H)
        return;
    const int hh = g.n >> 1;
    float diff[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
        diff[c] = 0.0f;
    for (int k = 0; k < K; ++k)
    {
        const short2 m = ((const short2*)fwd_maps.p[k])[(size_t)i * W + j];
        const int fx = m.x, fy = m.y;
        if ((unsigned)fx >= (unsigned)W || (unsigned)fy >= (unsigned)H)
            continue; // remap border: constant 0
        const char4* sg = (const char4*)signs.p[k];
        float a[CH], ra[CH], rb[CH];
        upsampled_blur_row<CH>(sg, W, scale, fy, fx, g, ra);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            a[c] = ra[c] * g.k[hh];
        for (int t = 1; t <= hh; ++t)
        {
            upsampled_blur_row<CH>(sg, W, scale, reflect101(fy - t, H), fx, g, ra);
            upsampled_blur_row<CH>(sg, W, scale, reflect101(fy + t, H), fx, g, rb);
#pragma unroll
            for (int c = 0; c < CH; ++c)
                a[c] += (ra[c] + rb[c]) * g.k[hh + t];
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
            diff[c] = diff[c] + a[c];
    }
    const float* P = X + ((size_t)i * W + j) * CH;
    float* D = X_out + ((size_t)i * W + j) * CH;
    if (lambda > 0)
    {
        float reg[CH];
        btv_at<CH>(X, H, W, i, j, bw, reg);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            diff[c] = (float)((double)diff[c] + (double)reg[c] * (-lambda));
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
        D[c] = (float)((double)P[c] + (double)diff[c] * tau);
}

// ---------------------------------------------------------------------------
template <int CH, bool TO8U>
__global__ void k_crop_convert(const float* __restrict__ X, int rows, int cols, int border, void* __restrict__ dst)
{
    const int oc = cols - 2 * border, orr = rows - 2 * border;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= oc || y >= orr)
        return;
    const float* P = X + ((size_t)(y + border) * cols + (x + border)) * CH;
    const size_t o = ((size_t)y * oc + x) * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c)
    {
        if (TO8U)
        {
            // convertTo(CV_8U): cvRound (half to even) then clamp; NaN -> 0
            float r = rintf(P[c]);
            r = fminf(fmaxf(r, 0.0f), 255.0f);
            ((uint8_t*)dst)[o + c] = (uint8_t)(int)r;
        }
        else
            ((float*)dst)[o + c] = P[c];
    }
}

__global__ void k_convert_8u_to_32f(const uint8_t* __restrict__ src, size_t count, float* __restrict__ dst)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count)
        dst[e] = (float)src[e];
}

inline dim3 grid2d(int cols, int rows, dim3 b, int z = 1)
{
    return dim3((cols + b.x - 1) / b.x, (rows + b.y - 1) / b.y, z);
}

int check_launch(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return hip_fail(e, what);
    return 0;
}

} // namespace

// ---------------------------------------------------------------------------
int launch_relative_motions(const PtrList& fwd, const MutPtrList& rel, int n, int base_idx, size_t count,
                            hipStream_t st)
{
    const int bs = 256;
    hipLaunchKernelGGL(k_relative_motions, dim3((unsigned)((count + bs - 1) / bs)), dim3(bs), 0, st, fwd, rel, n,
                       base_idx, count);
    return check_launch("relative_motions");
}

int launch_build_maps(const float* rel, int rows, int cols, int scale, const CubicTables& t, int16_t* fwd_map,
                      int16_t* bwd_map, hipStream_t st)
{
    const dim3 b(64, 4);
    hipLaunchKernelGGL(k_build_maps, grid2d(cols * scale, rows * scale, b), b, 0, st, rel, rows, cols, scale, t,
                       (short2*)fwd_map, (short2*)bwd_map);
    return check_launch("build_maps");
}

int launch_resize_cubic(const float* src, int srows, int scols, int ch, const CubicTables& t, float* dst, int drows,
                        int dcols, hipStream_t st)
{
    const dim3 b(64, 4);
    const dim3 g = grid2d(dcols, drows, b);
    switch (ch)
    {
    case 1: hipLaunchKernelGGL(k_resize_cubic<1>, g, b, 0, st, src, srows, scols, t, dst, drows, dcols); break;
    case 2: hipLaunchKernelGGL(k_resize_cubic<2>, g, b, 0, st, src, srows, scols, t, dst, drows, dcols); break;
    case 3: hipLaunchKernelGGL(k_resize_cubic<3>, g, b, 0, st, src, srows, scols, t, dst, drows, dcols); break;
    case 4: hipLaunchKernelGGL(k_resize_cubic<4>, g, b, 0, st, src, srows, scols, t, dst, drows, dcols); break;
    default: set_error("resize_cubic: channels must be 1..4"); return 1;
    }
    return check_launch("resize_cubic");
}

int launch_btv_regularization(const float* src, int rows, int cols, int ch, const BtvW& w, float* dst, hipStream_t st)
{
    const dim3 b(64, 4);
    const dim3 g = grid2d(cols, rows, b);
    switch (ch)
    {
    case 1: hipLaunchKernelGGL(k_btv_regularization<1>, g, b, 0, st, src, rows, cols, w, dst); break;
    case 3: hipLaunchKernelGGL(k_btv_regularization<3>, g, b, 0, st, src, rows, cols, w, dst); break;
    default: set_error("btv_regularization: channels must be 1 or 3"); return 1;
    }
    return check_launch("btv_regularization");
}

int launch_forward_sign(const float* X, int hr_rows, int hr_cols, int ch, const PtrList& bwd_maps, const PtrList& y,
                        const MutPtrList& signs, int K, int scale, const GaussK& g, hipStream_t st)
{
    const dim3 b(64, 4);
    const dim3 gr = grid2d(hr_cols / scale, hr_rows / scale, b, K);
    switch (ch)
    {
    case 1:
        hipLaunchKernelGGL(k_forward_sign<1>, gr, b, 0, st, X, hr_rows, hr_cols, bwd_maps, y, signs, scale, g);
        break;
    case 3:
        hipLaunchKernelGGL(k_forward_sign<3>, gr, b, 0, st, X, hr_rows, hr_cols, bwd_maps, y, signs, scale, g);
        break;
    default: set_error("forward_sign: channels must be 1 or 3"); return 1;
    }
    return check_launch("forward_sign");
}

int launch_backproject_update(const float* X, float* X_out, int hr_rows, int hr_cols, int ch, const PtrList& fwd_maps,
                              const PtrList& signs, int K, int scale, const GaussK& g, const BtvW& w, double tau,
                              double lambda, hipStream_t st)
{
    const dim3 b(64, 4);
    const dim3 gr = grid2d(hr_cols, hr_rows, b);
    switch (ch)
    {
    case 1:
        hipLaunchKernelGGL(k_backproject_update<1>, gr, b, 0, st, X, X_out, hr_rows, hr_cols, fwd_maps, signs, K,
                           scale, g, w, tau, lambda);
        break;
    case 3:
        hipLaunchKernelGGL(k_backproject_update<3>, gr, b, 0, st, X, X_out, hr_rows, hr_cols, fwd_maps, signs, K,
                           scale, g, w, tau, lambda);
        break;
    default: set_error("backproject_update: channels must be 1 or 3"); return 1;
    }
    return check_launch("backproject_update");
}

int launch_crop_convert(const float* X, int rows, int cols, int ch, int border, void* dst, int depth, hipStream_t st)
{
    const dim3 b(64, 4);
    const dim3 g = grid2d(cols - 2 * border, rows - 2 * border, b);
    const bool u8 = depth == 0;
    if (ch == 1 && u8)
        hipLaunchKernelGGL((k_crop_convert<1, true>), g, b, 0, st, X, rows, cols, border, dst);
    else if (ch == 1)
        hipLaunchKernelGGL((k_crop_convert<1, false>), g, b, 0, st, X, rows, cols, border, dst);
    else if (ch == 3 && u8)
        hipLaunchKernelGGL((k_crop_convert<3, true>), g, b, 0, st, X, rows, cols, border, dst);
    else if (ch == 3)
        hipLaunchKernelGGL((k_crop_convert<3, false>), g, b, 0, st, X, rows, cols, border, dst);
    else
    {
        set_error("crop_convert: channels must be 1 or 3");
        return 1;
    }
    return check_launch("crop_convert");
}

int launch_convert_8u_to_32f(const uint8_t* src, size_t count, float* dst, hipStream_t st)
{
    const int bs = 256;
    hipLaunchKernelGGL(k_convert_8u_to_32f, dim3((unsigned)((count + bs - 1) / bs)), dim3(bs), 0, st, src, count, dst);
    return check_launch("convert_8u_to_32f");
}

} // namespace srx
