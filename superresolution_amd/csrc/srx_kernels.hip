// HIP kernels of the BTV-L1 solver for gfx950 (MI355X).
//
// Build with -ffp-contract=off: the solver's sign() steps make the iteration
// sensitive to the last bit, so every float operation below rounds exactly as
// written, in the order of the OpenCV routines the reference calls.
//
// Data layout in HBM
//   images        float32, channels interleaved, dense rows.  Solver-owned HR
//                 images carry one extra all-zero pixel at index rows*cols
//                 (the value of cv::remap's constant border).
//   raw maps      int16 (x, y) per HR pixel: the source index that
//                 cv::remap(INTER_NEAREST) derives from its float map.
//   decoded maps  one uint32 per HR pixel, derived from the raw maps for the
//                 specialised kernels:
//                   backward: linear source pixel index, or rows*cols (the
//                             zero pixel) when the source is outside;
//                   forward:  see kFwd* below.
//   signs         one uint32 per LR pixel and window frame: sign+1 (0,1,2) of
//                 channel c in bits 10c..10c+1.
//
// Two families of kernels compute the same values: generic ones (any scale,
// blur size, BTV size) and specialised ones (scale 4, 5-tap blur, BTV 7).
#include "srx_common.h"

#include <cstring>

namespace srx
{

namespace
{

// forward decoded-map entry (scale 4, 5-tap blur): a target (fx, fy) touches at
// most the 2x2 block of LR sign cells whose top-left cell is `cell`
//   bits 0..23   cell index (row*w + col)
//   bits 24..28  weight-table index: 0..15 = (fx & 3) | (fy & 3) << 2,
//                16 = target outside the image (all weights 0),
//                17 = target within 2 px of the image edge, where reflect101
//                     comes into play (weights 0 here; evaluated generically)
constexpr uint32_t kFwdCellMask = 0x00FFFFFFu;
constexpr int kFwdLutShift = 24;
constexpr uint32_t kFwdLutOutside = 16, kFwdLutSlow = 17;
constexpr int kFwdLutSize = 18;

typedef int v4i __attribute__((ext_vector_type(4)));
// volatile LDS vector access keeps 16-byte ds_read_b128 from being split into narrower reads
typedef __attribute__((address_space(3))) const volatile v4i lds_cv4i;

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

// Blocks are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).
// Give each XCD one contiguous range of the work list so that neighbouring
// work items (which share halos and gather targets) meet in one L2.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned b, unsigned n)
{
    const unsigned xcd = b & 7u, local = b >> 3;
    const unsigned q = n >> 3, r = n & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
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

// decoded map entries (layout at the top of this file); w = C / 4 LR columns
__device__ __forceinline__ uint32_t decode_bwd(int bx, int by, int R, int C)
{
    return ((unsigned)bx < (unsigned)C && (unsigned)by < (unsigned)R) ? (uint32_t)(by * C + bx) : (uint32_t)(R * C);
}
__device__ __forceinline__ uint32_t decode_fwd(int fx, int fy, int R, int C)
{
    if ((unsigned)fx >= (unsigned)C || (unsigned)fy >= (unsigned)R)
        return kFwdLutOutside << kFwdLutShift;
    if (fx < 2 || fx > C - 3 || fy < 2 || fy > R - 3)
        return kFwdLutSlow << kFwdLutShift;
    const int px = fx & 3, py = fy & 3;
    const int col = (fx >> 2) + (px == 3), row = (fy >> 2) + (py == 3);
    return (uint32_t)(row * (C >> 2) + col) | ((uint32_t)(px | (py << 2)) << kFwdLutShift);
}

// upscaleMotions + buildMotionMaps + remap's float->short index conversion
// (lib/src/btv_l1.cpp:59-101; cv::remap INTER_NEAREST), plus the decoded maps
// for the specialised kernels when dec_fwd / dec_bwd are given (scale 4 only).
__global__ void k_build_maps(const float* __restrict__ rel, int rows, int cols, int scale, CubicTables t,
                             short2* __restrict__ fwd_map, short2* __restrict__ bwd_map, uint32_t* __restrict__ dec_fwd,
                             uint32_t* __restrict__ dec_bwd)
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
    const int fx = round_sat16((float)x - mx), fy = round_sat16((float)y - my);
    const int bx = round_sat16((float)x + mx), by = round_sat16((float)y + my);
    const size_t o = (size_t)y * C + x;
    fwd_map[o] = make_short2((short)fx, (short)fy);
    bwd_map[o] = make_short2((short)bx, (short)by);
    if (dec_bwd)
        dec_bwd[o] = decode_bwd(bx, by, R, C);
    if (dec_fwd)
        dec_fwd[o] = decode_fwd(fx, fy, R, C);
}

// raw int16 maps -> decoded maps (same rules as k_build_maps), for callers that hold raw maps only
__global__ void k_decode_maps(const short2* __restrict__ fwd_map, const short2* __restrict__ bwd_map, int R, int C,
                              uint32_t* __restrict__ dec_fwd, uint32_t* __restrict__ dec_bwd)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= C || y >= R)
        return;
    const size_t o = (size_t)y * C + x;
    if (dec_bwd)
        dec_bwd[o] = decode_bwd(bwd_map[o].x, bwd_map[o].y, R, C);
    if (dec_fwd)
        dec_fwd[o] = decode_fwd(fwd_map[o].x, fwd_map[o].y, R, C);
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

// ===========================================================================
// Generic kernels
// ===========================================================================

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

// sign(y - D H M X) at one LR pixel, any scale / blur size (lib/src/btv_l1.cpp:428-434)
template <int CH>
__device__ __forceinline__ uint32_t forward_sign_generic(const float* __restrict__ X, int H, int W,
                                                         const short2* __restrict__ map,
                                                         const float* __restrict__ yk, int x, int y, int scale,
                                                         const GaussK& g)
{
    const int w = W / scale;
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
    const float* yp = yk + ((size_t)y * w + x) * CH;
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c)
        s |= (uint32_t)((int)dsign(yp[c], acc[c]) + 1) << (10 * c);
    return s;
}

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
    ((uint32_t*)signs.p[k])[(size_t)y * w + x] =
        forward_sign_generic<CH>(X, H, W, (const short2*)bwd_maps.p[k], (const float*)ys.p[k], x, y, scale, g);
}

// Transposed chain.  B(fy, fx) = gauss(zero_insert(sign))(fy, fx) evaluated
// lazily: the zero-inserted image is non-zero only where both coordinates are
// multiples of scale.
template <int CH>
__device__ __forceinline__ void sign_at(const uint32_t* __restrict__ sg, int w, int scale, int Yr, int Xr,
                                        float (&o)[CH])
{
    // caller guarantees Yr % scale == 0
    if (Xr % scale == 0)
    {
        const uint32_t s = sg[(size_t)(Yr / scale) * w + (Xr / scale)];
#pragma unroll
        for (int c = 0; c < CH; ++c)
            o[c] = (float)((int)((s >> (10 * c)) & 3u) - 1);
    }
    else
    {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            o[c] = 0.0f;
    }
}

template <int CH>
__device__ __forceinline__ void upsampled_blur_row(const uint32_t* __restrict__ sg, int W, int scale, int Yr, int fx,
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

// (M^T H^T D^T sign)(i, j) for one frame: gather at the forward-map target
template <int CH>
__device__ __forceinline__ void backproject_generic(const uint32_t* __restrict__ sg, int H, int W, int scale, int fx,
                                                    int fy, const GaussK& g, float (&a)[CH])
{
    const int hh = g.n >> 1;
    float ra[CH], rb[CH];
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
        float a[CH];
        backproject_generic<CH>((const uint32_t*)signs.p[k], H, W, scale, fx, fy, g, a);
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

// ===========================================================================
// Specialised kernels: scale 4, 5-tap blur (and BTV kernel size 7 for the update)
// ===========================================================================

// value of lane-1 / lane+1 within the 16-lane DPP row (0 at the row's ends)
__device__ __forceinline__ float dpp_from_left(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111 /* row_shr:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_from_right(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101 /* row_shl:1 */, 0xF, 0xF, true));
}

// Forward chain sign(y_k - D H M_k X), one wave per (strip of 28 LR columns,
// group of LR rows, window frame).
//
// Lanes own HR columns, not LR pixels: lane p of a 16-lane DPP row holds the
// warped pixels of two adjacent HR columns (E = even, O = odd register), so a
// wave's map reads and X gathers run over consecutive addresses.  The 5-tap row
// pass at the even column of every odd lane is
//     E*k0 + (O[p-1] + O[p])*k1 + (E[p-1] + E[p+1])*k2
// with the neighbours fetched by DPP row shifts; 7 LR pixels per DPP row, 28
// per wave.  The wave walks down its LR rows keeping the five row-pass results
// of the current LR row in registers (four new HR rows per LR row).
// Work items are ordered frame-fastest so that the waves gathering from one
// region of X run together on one CU / XCD.
constexpr int kFwdRowsPerWave = 6;
constexpr int kFwdStrip = 28;

template <int CH>
__global__ __launch_bounds__(256) void k_forward_sign_s4(const float* __restrict__ X, int H, int W, PtrList dec_bwd,
                                                          PtrList ys, MutPtrList signs, int K, GaussK g,
                                                          int strips_x, unsigned n_items, unsigned n_blocks)
{
    const unsigned item = xcd_contiguous(blockIdx.x, n_blocks) * 4 + (threadIdx.x >> 6);
    if (item >= n_items)
        return;
    const int k = item % K;
    const unsigned tile = item / K;
    const int strip = tile % strips_x, rg = tile / strips_x;
    const int h = H >> 2, w = W >> 2;
    const int lane = threadIdx.x & 63, j = lane >> 4, q = lane & 15;
    const int x0 = strip * kFwdStrip, y0 = rg * kFwdRowsPerWave;
    const int y1 = min(y0 + kFwdRowsPerWave, h);

    const int col0 = 4 * x0 - 2 + 28 * j + 2 * q; // even register's HR column; the odd register holds col0 + 1
    const int cA = reflect101(col0, W), cB = reflect101(col0 + 1, W);
    const bool pair = (cB == cA + 1);             // plain interior lane: one 8-byte map load
    const int x = x0 + 7 * j + (q >> 1);          // LR pixel centred on col0 (odd lanes 1..13)
    const bool owner = (q & 1) && q <= 13 && x < w;

    const uint32_t* map = (const uint32_t*)dec_bwd.p[k];
    const float k0 = g.k[2], k1 = g.k[3], k2 = g.k[4];

    auto row_pass = [&](int Y, float (&out)[CH]) {
        const uint32_t* mr = map + (size_t)reflect101(Y, H) * W;
        uint32_t ia, ib;
        if (pair)
        {
            const uint2 m2 = *(const uint2*)(mr + cA);
            ia = m2.x, ib = m2.y;
        }
        else
        {
            ia = mr[cA];
            ib = mr[cB];
        }
        const float* pe = X + (size_t)ia * CH;
        const float* po = X + (size_t)ib * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c)
        {
            const float e = pe[c], o = po[c];
            const float u = dpp_from_left(o) + o;                  // columns col0-1, col0+1
            const float v = dpp_from_left(e) + dpp_from_right(e);  // columns col0-2, col0+2
            float s = e * k0;
            s += u * k1;
            s += v * k2;
            out[c] = s;
        }
    };

    float rr[5][CH];
    row_pass(4 * y0 - 2, rr[0]);
    for (int y = y0; y < y1; ++y)
    {
#pragma unroll
        for (int r = 1; r < 5; ++r)
            row_pass(4 * y - 2 + r, rr[r]);
        if (owner)
        {
            const float* yp = (const float*)ys.p[k] + ((size_t)y * w + x) * CH;
            uint32_t sg = 0;
#pragma unroll
            for (int c = 0; c < CH; ++c)
            {
                float s = rr[2][c] * k0;
                s += (rr[1][c] + rr[3][c]) * k1;
                s += (rr[0][c] + rr[4][c]) * k2;
                sg |= (uint32_t)((int)dsign(yp[c], s) + 1) << (10 * c);
            }
            ((uint32_t*)signs.p[k])[(size_t)y * w + x] = sg;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
            rr[0][c] = rr[4][c];
    }
}

// Integer weights of the transposed 5-tap blur for dyadic kernels (k*16 integer):
// per weight-table index the weights of the 2x2 block of sign cells, in units of
// 1/256, and their sum (entries 16 and 17 are all zero).
struct FastW
{
    uint32_t w[kFwdLutSize][4]; // [AA, AB, BA, BB]
    uint32_t wsum[kFwdLutSize];
};

// order-preserving float <-> int key (an involution); -0 is folded into +0 so
// that integer order equals float order for all non-NaN values
__device__ __forceinline__ int float_key(float x)
{
    const int b = __float_as_int(x + 0.0f);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float key_float(int k)
{
    return __int_as_float(k ^ ((k >> 31) & 0x7fffffff));
}
// sign(a - b) for keys: saturating subtract, then clamp to [-1, 1]
__device__ __forceinline__ int key_sign(int a, int b)
{
    int d, s;
    asm("v_sub_i32 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(s) : "v"(d));
    return s;
}

// Back-projection of all K frames + BTV regulariser + update for one 64x16 tile
// of HR pixels per block; each thread owns 4 consecutive pixels of one row.
//  * per frame: one 16-byte load of 4 decoded forward-map entries (fetched one
//    frame ahead), then per pixel the 2x2 block of sign cells times the entry's
//    integer weights.  Exact for dyadic blur kernels: every partial sum is a
//    multiple of 1/256.  The loop is branch-free (entries outside the image carry
//    zero weights) so that all gathers of a frame are in flight together; the
//    rare targets next to the image edge are added by a second, generic pass.
//  * BTV: the tile of X (+3 px halo) sits in LDS as order-preserving integer
//    keys; sign(a-b) is a saturating subtract and a clamp; accumulation order is
//    the reference's (lib/src/btv_l1.cpp:221-233).  NaN pixels are ordered as
//    large values instead of comparing unordered.
#ifndef SRX_B_WAVES
#define SRX_B_WAVES 4
#endif
template <int CH>
__global__ __launch_bounds__(256, SRX_B_WAVES) void k_backproject_update_s4(const float* __restrict__ X, float* __restrict__ X_out,
                                                                int H, int W, PtrList dec_fwd, PtrList raw_fwd,
                                                                PtrList signs, int K, GaussK g, FastW fw, BtvW bw,
                                                                double tau, double lambda, int tiles_x,
                                                                unsigned n_tiles)
{
    constexpr int TXN = 16, TYN = 16, PXT = 4;
    constexpr int TW = TXN * PXT;       // 64 px
    constexpr int HALO = 3, HL = 4;     // halo; left halo padded to 4 px so that row loads are 16-byte aligned
    constexpr int LW = TW + HL + 4;     // 72 px held per LDS row: [j0-4, j0+68)
    constexpr int LROWS = TYN + 2 * HALO;
    constexpr int PITCH = CH == 3 ? 256 : 128; // floats; a multiple of 64 keeps ds_read_b128 conflict-free
    constexpr int NV = CH == 3 ? 9 : 3;        // 16-byte LDS reads per window row
    constexpr int OFF = CH == 3 ? 3 : 1;       // first needed float within the aligned window
    constexpr int NE = PXT * CH;               // outputs per thread
    __shared__ __attribute__((aligned(16))) int tile[LROWS * PITCH];
    __shared__ uint4 lutw[kFwdLutSize];
    __shared__ uint32_t luts[kFwdLutSize];

    const unsigned t = xcd_contiguous(blockIdx.x, n_tiles);
    if (t >= n_tiles)
        return;
    const int j0 = (t % tiles_x) * TW, i0 = (t / tiles_x) * TYN;
    const int tid = threadIdx.x;
    const int tx = tid & (TXN - 1), ty = tid >> 4;
    const int w = W >> 2;
    const int i = min(i0 + ty, H - 1), j = min(j0 + tx * PXT, W - PXT); // clamped; stores are masked below
    const bool live = (i0 + ty) < H && (j0 + tx * PXT) < W; // W is a multiple of 4: a thread's pixels are all in or all out
    const size_t pix = (size_t)i * W + j;

    // first frame's map entries: issued before the tile is staged
    uint4 cur = make_uint4(0, 0, 0, 0);
    if (K > 0)
        cur = *(const uint4*)((const uint32_t*)dec_fwd.p[0] + pix);

    if (tid < kFwdLutSize)
    {
        lutw[tid] = make_uint4(fw.w[tid][0], fw.w[tid][1], fw.w[tid][2], fw.w[tid][3]);
        luts[tid] = fw.wsum[tid];
    }
    // ---- stage the X tile (as keys) in LDS; all loads are issued before the first is consumed
    {
        constexpr int V4 = LW * CH / 4; // float4 per row: 54 or 18
        constexpr int NIT = (LROWS * V4 + 255) / 256;
        const int rowf = W * CH;
        float4 stage[NIT];
        bool have[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it)
        {
            const int idx = tid + it * 256;
            const int r = idx / V4, c4 = idx - r * V4;
            const int gi = i0 - HALO + r;
            const int f0 = (j0 - HL) * CH + 4 * c4;
            have[it] = idx < LROWS * V4 && gi >= 0 && gi < H && f0 >= 0 && f0 + 3 < rowf;
            stage[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (have[it])
                stage[it] = *(const float4*)(X + (size_t)gi * rowf + f0);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it)
        {
            const int idx = tid + it * 256;
            const int r = idx / V4, c4 = idx - r * V4;
            if (idx < LROWS * V4)
            {
                const float4 v = stage[it];
                // outside the image the tile holds key(0); those cells are never read by an interior pixel
                *(v4i*)&tile[r * PITCH + 4 * c4] = v4i{float_key(v.x), float_key(v.y), float_key(v.z), float_key(v.w)};
            }
        }
    }
    __syncthreads();

    // ---- BTV regulariser from the LDS tile
    float reg[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e)
        reg[e] = 0.0f;
    const int lrow = ty + HALO;
    lds_cv4i* tp = (lds_cv4i*)tile + (lrow * PITCH + tx * PXT * CH) / 4; // aligned window start of the centre row
    constexpr int RV = PITCH / 4;                                        // vectors per LDS row
    v4i cv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v)
        cv[v] = tp[v];
    const int* cw = (const int*)cv;
    int ctr[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e)
        ctr[e] = cw[OFF + HALO * CH + e];
    if (lambda > 0)
    {
        int ind = 0;
#pragma unroll
        for (int m = 0; m <= 3; ++m)
        {
            v4i pv[NV], mv[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v)
            {
                pv[v] = m == 0 ? cv[v] : tp[m * RV + v];
                mv[v] = m == 0 ? cv[v] : tp[-m * RV + v];
            }
            const int* pw = (const int*)pv;
            const int* mw = (const int*)mv;
#pragma unroll
            for (int l = 3; l + m >= 0; --l, ++ind)
            {
                if (m == 0 && l == 0)
                    continue; // both signs compare the pixel with itself: adds w * 0
                const float wt = bw.w[ind];
#pragma unroll
                for (int e = 0; e < NE; ++e)
                {
                    const int c0 = OFF + HALO * CH + e;
                    const int d = key_sign(ctr[e], pw[c0 + CH * l]) + key_sign(ctr[e], mw[c0 - CH * l]);
                    // w * d is exact (d in -2..2), so the fused form rounds once, like acc += w * d
                    reg[e] = __builtin_fmaf(wt, (float)d, reg[e]);
                }
            }
            // keep the next row pair's LDS reads behind this pair's arithmetic: hoisting them
            // would hold several 36-register windows live at once
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- sum over frames of M^T H^T D^T sign, in units of 1/256
    int acc[PXT][CH];
    int bias[PXT];
    uint32_t slow = 0;
#pragma unroll
    for (int p = 0; p < PXT; ++p)
    {
        bias[p] = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            acc[p][c] = 0;
    }
#ifndef SRX_B_UNROLL
#define SRX_B_UNROLL 1
#endif
    // the next frame's map entries are fetched while the current frame's sign cells are gathered
#pragma unroll SRX_B_UNROLL
    for (int k = 0; k < K; ++k)
    {
        uint4 nxt = cur;
        if (k + 1 < K)
            nxt = *(const uint4*)((const uint32_t*)dec_fwd.p[k + 1] + pix);
        const uint32_t* sg = (const uint32_t*)signs.p[k];
        const uint32_t ev[PXT] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (int p = 0; p < PXT; ++p)
        {
            const uint32_t e = ev[p];
            const uint32_t* s0 = sg + (e & kFwdCellMask);
            const uint32_t li = e >> kFwdLutShift;
            const uint4 wq = lutw[li];
            // operands fit 24 bits: full-rate v_mad_u32_u24
            uint32_t v = __umul24(s0[0], wq.x);
            v += __umul24(s0[1], wq.y);
            v += __umul24(s0[w], wq.z);
            v += __umul24(s0[w + 1], wq.w);
            bias[p] += luts[li];
            slow |= (li == kFwdLutSlow) ? 1u : 0u;
#pragma unroll
            for (int c = 0; c < CH; ++c)
                acc[p][c] += (int)((v >> (10 * c)) & 1023u);
        }
        cur = nxt;
    }
    if (slow)
    {
        // targets within 2 px of the image edge: generic evaluation with reflect101
        for (int k = 0; k < K; ++k)
        {
            const uint32_t* sg = (const uint32_t*)signs.p[k];
            const uint32_t* de = (const uint32_t*)dec_fwd.p[k] + pix;
#pragma unroll 1
            for (int pp = 0; pp < PXT; ++pp)
            {
                if ((de[pp] >> kFwdLutShift) != kFwdLutSlow)
                    continue;
                const short2 m = ((const short2*)raw_fwd.p[k])[pix + pp];
                float a[CH];
                backproject_generic<CH>(sg, H, W, 4, m.x, m.y, g, a);
#pragma unroll
                for (int p = 0; p < PXT; ++p)
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        acc[p][c] += p == pp ? (int)(a[c] * 256.0f) : 0;
            }
        }
    }

    // ---- update
    if (!live)
        return;
    float outv[NE];
#pragma unroll
    for (int p = 0; p < PXT; ++p)
    {
        const bool interior = i >= HALO && i < H - HALO && (j + p) >= HALO && (j + p) < W - HALO;
#pragma unroll
        for (int c = 0; c < CH; ++c)
        {
            const int e = p * CH + c;
            float d = (float)(acc[p][c] - bias[p]) * 0.00390625f;
            if (lambda > 0)
            {
                const float rg = interior ? reg[e] : 0.0f;
                d = (float)((double)d + (double)rg * (-lambda));
            }
            outv[e] = (float)((double)key_float(ctr[e]) + (double)d * tau);
        }
    }
    float4* D = (float4*)(X_out + pix * CH);
#pragma unroll
    for (int v = 0; v < NE / 4; ++v)
        D[v] = make_float4(outv[4 * v], outv[4 * v + 1], outv[4 * v + 2], outv[4 * v + 3]);
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

// round the block count up to a multiple of 8 so that every XCD's range exists
inline unsigned xcd_grid(unsigned n)
{
    return (n + 7u) & ~7u;
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
                      int16_t* bwd_map, uint32_t* dec_fwd, uint32_t* dec_bwd, hipStream_t st)
{
    if ((dec_fwd || dec_bwd) && scale != 4)
    {
        set_error("build_maps: decoded maps exist for scale 4 only");
        return 1;
    }
    const dim3 b(64, 4);
    hipLaunchKernelGGL(k_build_maps, grid2d(cols * scale, rows * scale, b), b, 0, st, rel, rows, cols, scale, t,
                       (short2*)fwd_map, (short2*)bwd_map, dec_fwd, dec_bwd);
    return check_launch("build_maps");
}

int launch_decode_maps(const int16_t* fwd_map, const int16_t* bwd_map, int hr_rows, int hr_cols, uint32_t* dec_fwd,
                       uint32_t* dec_bwd, hipStream_t st)
{
    const dim3 b(64, 4);
    hipLaunchKernelGGL(k_decode_maps, grid2d(hr_cols, hr_rows, b), b, 0, st, (const short2*)fwd_map,
                       (const short2*)bwd_map, hr_rows, hr_cols, dec_fwd, dec_bwd);
    return check_launch("decode_maps");
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

bool forward_fast_eligible(int hr_rows, int hr_cols, int ch, int scale, const GaussK& g)
{
    return scale == 4 && g.n == 5 && (ch == 1 || ch == 3) && hr_rows % 4 == 0 && hr_cols % 4 == 0 &&
           (size_t)hr_rows * hr_cols < (1u << 30);
}

int launch_forward_sign(const float* X, int hr_rows, int hr_cols, int ch, const PtrList& raw_bwd,
                        const PtrList* dec_bwd, const PtrList& y, const MutPtrList& signs, int K, int scale,
                        const GaussK& g, hipStream_t st)
{
    if (ch != 1 && ch != 3)
    {
        set_error("forward_sign: channels must be 1 or 3");
        return 1;
    }
    if (dec_bwd && forward_fast_eligible(hr_rows, hr_cols, ch, scale, g))
    {
        const int w = hr_cols / 4, h = hr_rows / 4;
        const int strips_x = (w + kFwdStrip - 1) / kFwdStrip;
        const int row_groups = (h + kFwdRowsPerWave - 1) / kFwdRowsPerWave;
        const unsigned n_items = (unsigned)strips_x * row_groups * K; // one wave each
        const unsigned n_blocks = (n_items + 3) / 4;
        const dim3 gr(xcd_grid(n_blocks)), b(256);
        if (ch == 1)
            hipLaunchKernelGGL(k_forward_sign_s4<1>, gr, b, 0, st, X, hr_rows, hr_cols, *dec_bwd, y, signs, K, g,
                               strips_x, n_items, n_blocks);
        else
            hipLaunchKernelGGL(k_forward_sign_s4<3>, gr, b, 0, st, X, hr_rows, hr_cols, *dec_bwd, y, signs, K, g,
                               strips_x, n_items, n_blocks);
        return check_launch("forward_sign_s4");
    }
    const dim3 b(64, 4);
    const dim3 gr = grid2d(hr_cols / scale, hr_rows / scale, b, K);
    if (ch == 1)
        hipLaunchKernelGGL(k_forward_sign<1>, gr, b, 0, st, X, hr_rows, hr_cols, raw_bwd, y, signs, scale, g);
    else
        hipLaunchKernelGGL(k_forward_sign<3>, gr, b, 0, st, X, hr_rows, hr_cols, raw_bwd, y, signs, scale, g);
    return check_launch("forward_sign");
}

// The integer path needs every blur weight to be a multiple of 1/16 (true for
// the sigma = 0 table {1,4,6,4,1}/16).
static bool make_fast_weights(const GaussK& g, FastW& fw)
{
    if (g.n != 5)
        return false;
    int ki[3];
    for (int i = 0; i < 3; ++i)
    {
        const float v = g.k[2 + i] * 16.0f;
        if (!(v >= 0.0f && v <= 15.0f) || v != (float)(int)v || g.k[2 + i] != g.k[2 - i])
            return false;
        ki[i] = (int)v;
    }
    // per phase p = f & 3: slot A carries tap j = {0,-1,-2,+1}[p], slot B only exists for p == 2 (tap +2)
    const int gA[4] = {ki[0], ki[1], ki[2], ki[1]};
    const int gB[4] = {0, 0, ki[2], 0};
    std::memset(&fw, 0, sizeof(fw));
    for (int py = 0; py < 4; ++py)
        for (int px = 0; px < 4; ++px)
        {
            const int p = px | (py << 2);
            fw.w[p][0] = gA[px] * gA[py];
            fw.w[p][1] = gB[px] * gA[py];
            fw.w[p][2] = gA[px] * gB[py];
            fw.w[p][3] = gB[px] * gB[py];
            fw.wsum[p] = fw.w[p][0] + fw.w[p][1] + fw.w[p][2] + fw.w[p][3];
            if (2 * fw.wsum[p] > 1023u)
                return false; // 10-bit fields
        }
    return true;
}

bool backproject_fast_eligible(int hr_rows, int hr_cols, int ch, int scale, const GaussK& g, const BtvW& w)
{
    FastW fw;
    return scale == 4 && w.radius == 3 && (ch == 1 || ch == 3) && hr_rows % 4 == 0 && hr_cols % 4 == 0 &&
           (size_t)(hr_rows / 4) * (hr_cols / 4) < (1u << 24) && make_fast_weights(g, fw);
}

int launch_backproject_update(const float* X, float* X_out, int hr_rows, int hr_cols, int ch, const PtrList& raw_fwd,
                              const PtrList* dec_fwd, const PtrList& signs, int K, int scale, const GaussK& g,
                              const BtvW& w, double tau, double lambda, hipStream_t st)
{
    if (ch != 1 && ch != 3)
    {
        set_error("backproject_update: channels must be 1 or 3");
        return 1;
    }
    FastW fw;
    if (dec_fwd && backproject_fast_eligible(hr_rows, hr_cols, ch, scale, g, w) && make_fast_weights(g, fw))
    {
        const int tiles_x = (hr_cols + 63) / 64, tiles_y = (hr_rows + 15) / 16;
        const unsigned n_tiles = (unsigned)tiles_x * tiles_y;
        const dim3 gr(xcd_grid(n_tiles)), b(256);
        if (ch == 1)
            hipLaunchKernelGGL(k_backproject_update_s4<1>, gr, b, 0, st, X, X_out, hr_rows, hr_cols, *dec_fwd, raw_fwd,
                               signs, K, g, fw, w, tau, lambda, tiles_x, n_tiles);
        else
            hipLaunchKernelGGL(k_backproject_update_s4<3>, gr, b, 0, st, X, X_out, hr_rows, hr_cols, *dec_fwd, raw_fwd,
                               signs, K, g, fw, w, tau, lambda, tiles_x, n_tiles);
        return check_launch("backproject_update_s4");
    }
    const dim3 b(64, 4);
    const dim3 gr = grid2d(hr_cols, hr_rows, b);
    if (ch == 1)
        hipLaunchKernelGGL(k_backproject_update<1>, gr, b, 0, st, X, X_out, hr_rows, hr_cols, raw_fwd, signs, K, scale,
                           g, w, tau, lambda);
    else
        hipLaunchKernelGGL(k_backproject_update<3>, gr, b, 0, st, X, X_out, hr_rows, hr_cols, raw_fwd, signs, K, scale,
                           g, w, tau, lambda);
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
