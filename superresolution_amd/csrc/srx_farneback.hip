// Built-in Farneback dense optical flow for gfx950 (polynomial expansion; the reference's default
// flow class wraps OpenCV's calcOpticalFlowFarneback, lib/src/optical_flow.cpp:84-87).
// Every float / double operation follows the order of the CPU statement of this flow used by this
// repository's tests; the file is built with -ffp-contract=off.
#include "srx_farneback.h"
#include "srx_tvl1.h"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace srx
{

namespace
{
constexpr int kMaxTaps = 127; // blur ahead of the resize to a pyramid level
constexpr int kMaxPoly = 16;

struct FbTaps
{
    float k[kMaxTaps + 1];
    int n;
};
struct FbPoly
{
    float g[kMaxPoly + 1], xg[kMaxPoly + 1], xxg[kMaxPoly + 1];
    double ig11, ig03, ig33, ig55;
    int n;
};

int fb_bad(const std::string& msg)
{
    set_error(msg);
    return SRX_ERR_INVALID;
}

// cv::getGaussianKernel(n, sigma, CV_32F)
void gaussian_taps(int n, double sigma, float* k)
{
    static const float k3[3] = {0.25f, 0.5f, 0.25f};
    const bool fixed = n == 3 && sigma <= 0;
    const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; ++i)
    {
        const double x = i - (n - 1) * 0.5;
        const double t = fixed ? (double)k3[i] : std::exp(scale2X * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i)
        k[i] = (float)(k[i] * sum);
}

// blur of the full-size image ahead of the resize to the level of this scale
bool smooth_taps(double scale, FbTaps& t)
{
    const double sigma = (1. / scale - 1) * 0.5;
    int n = (int)std::nearbyint(sigma * 5) | 1;
    n = std::max(n, 3);
    if (n > kMaxTaps)
        return false;
    t.n = n;
    gaussian_taps(n, sigma, t.k);
    return true;
}

// taps of the expansion and the entries of the inverse moment matrix it uses
void poly_setup(int n, double sigma, FbPoly& q)
{
    float gf[2 * kMaxPoly + 1];
    if (sigma < FLT_EPSILON)
        sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; ++x)
    {
        gf[x + n] = (float)std::exp(-x * x / (2 * sigma * sigma));
        s += gf[x + n];
    }
    s = 1. / s;
    for (int x = -n; x <= n; ++x)
        gf[x + n] = (float)(gf[x + n] * s);
    for (int x = 0; x <= n; ++x)
    {
        q.g[x] = gf[x + n];
        q.xg[x] = (float)(x * gf[x + n]);
        q.xxg[x] = (float)(x * x * gf[x + n]);
    }
    double G00 = 0, G11 = 0, G33 = 0, G55 = 0;
    for (int y = -n; y <= n; ++y)
        for (int x = -n; x <= n; ++x)
        {
            G00 += gf[y + n] * gf[x + n];
            G11 += gf[y + n] * gf[x + n] * x * x;
            G33 += gf[y + n] * gf[x + n] * x * x * x * x;
            G55 += gf[y + n] * gf[x + n] * x * x * y * y;
        }
    // inverse of the block [G00 G11 G11; G11 G33 G55; G11 G55 G33] in closed form
    const double a = G00, b = G11, c = G33, d = G55;
    const double den = a * (c + d) - 2 * b * b;
    q.ig11 = 1. / G11;
    q.ig03 = -b / den;
    q.ig33 = (a * c - b * b) / ((c - d) * den);
    q.ig55 = 1. / G55;
    q.n = n;
}

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
    return v < lo ? lo : v > hi ? hi : v;
}

// Gaussian blur, row pass: taps summed symmetrically for sizes <= 5, left to right otherwise
__global__ void k_fb_blur_rows(const float* __restrict__ src, int rows, int cols, FbTaps t, float* __restrict__ dst)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows)
        return;
    const float* S = src + (size_t)y * cols;
    const int n = t.n, h = n / 2;
    float s;
    if (n <= 5)
    {
        s = S[x] * t.k[h];
        for (int j = 1; j <= h; ++j)
            s += (S[reflect101(x - j, cols)] + S[reflect101(x + j, cols)]) * t.k[h + j];
    }
    else
    {
        s = t.k[0] * S[reflect101(x - h, cols)];
        for (int j = 1; j < n; ++j)
            s += t.k[j] * S[reflect101(x - h + j, cols)];
    }
    dst[(size_t)y * cols + x] = s;
}

__global__ void k_fb_blur_cols(const float* __restrict__ src, int rows, int cols, FbTaps t, float* __restrict__ dst)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows)
        return;
    const int h = t.n / 2;
    float s = src[(size_t)y * cols + x] * t.k[h];
    for (int j = 1; j <= h; ++j)
        s += (src[(size_t)reflect101(y - j, rows) * cols + x] + src[(size_t)reflect101(y + j, rows) * cols + x]) *
             t.k[h + j];
    dst[(size_t)y * cols + x] = s;
}

// cv::resize(INTER_LINEAR) of an image with CH interleaved channels; with mul != 0 every result
// is multiplied by it in double (the flow of the coarser level, times 1 / pyrScale)
template <int CH>
__global__ void k_fb_resize_linear(const float* __restrict__ src, int srows, int scols, float* __restrict__ dst,
                                   int drows, int dcols, double sx_, double sy_, double mul)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y * blockDim.y + threadIdx.y;
    if (dx >= dcols || dy >= drows)
        return;
    float fx = (float)((dx + 0.5) * sx_ - 0.5);
    int sx = (int)floor((double)fx);
    fx -= sx;
    if (sx < 0)
        fx = 0, sx = 0;
    if (sx >= scols - 1)
        fx = 0, sx = scols - 1;
    const bool last = sx >= scols - 1;
    const float a0 = 1.f - fx, a1 = fx;
    float fy = (float)((dy + 0.5) * sy_ - 0.5);
    const int sy = (int)floor((double)fy);
    fy -= sy;
    const float b0 = 1.f - fy, b1 = fy;
    const float* R0 = src + (size_t)clampi(sy, 0, srows - 1) * scols * CH;
    const float* R1 = src + (size_t)clampi(sy + 1, 0, srows - 1) * scols * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c)
    {
        const float h0 = last ? R0[(scols - 1) * CH + c] : R0[sx * CH + c] * a0 + R0[(sx + 1) * CH + c] * a1;
        const float h1 = last ? R1[(scols - 1) * CH + c] : R1[sx * CH + c] * a0 + R1[(sx + 1) * CH + c] * a1;
        const float v = h0 * b0 + h1 * b1;
        dst[((size_t)dy * dcols + dx) * CH + c] = mul != 0 ? (float)(v * mul) : v;
    }
}

// polynomial expansion, vertical pass: three running sums per pixel, float, replicated rows
__global__ void k_fb_poly_rows(const float* __restrict__ src, int rows, int cols, FbPoly q, float* __restrict__ row)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows)
        return;
    float t0 = src[(size_t)y * cols + x] * q.g[0], t1 = 0.f, t2 = 0.f;
    for (int k = 1; k <= q.n; ++k)
    {
        const float s0 = src[(size_t)clampi(y - k, 0, rows - 1) * cols + x];
        const float s1 = src[(size_t)clampi(y + k, 0, rows - 1) * cols + x];
        const float p = s0 + s1;
        t0 = t0 + q.g[k] * p;
        t1 = t1 + q.xg[k] * (s1 - s0);
        t2 = t2 + q.xxg[k] * p;
    }
    float* D = row + ((size_t)y * cols + x) * 3;
    D[0] = t0, D[1] = t1, D[2] = t2;
}

// horizontal pass, accumulated in double, and the five coefficients (y, x, yy, xx, xy)
__global__ void k_fb_poly_cols(const float* __restrict__ row, int rows, int cols, FbPoly q, float* __restrict__ dst)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows)
        return;
    const float* r = row + (size_t)y * cols * 3;
    double b1 = r[x * 3] * q.g[0], b2 = 0, b3 = r[x * 3 + 1] * q.g[0], b4 = 0, b5 = r[x * 3 + 2] * q.g[0], b6 = 0;
    for (int k = 1; k <= q.n; ++k)
    {
        const float* rp = r + clampi(x + k, 0, cols - 1) * 3;
        const float* rm = r + clampi(x - k, 0, cols - 1) * 3;
        const double tg = rp[0] + rm[0];
        b1 += tg * q.g[k];
        b4 += tg * q.xxg[k];
        b2 += (rp[0] - rm[0]) * q.xg[k];
        b3 += (rp[1] + rm[1]) * q.g[k];
        b6 += (rp[1] - rm[1]) * q.xg[k];
        b5 += (rp[2] + rm[2]) * q.g[k];
    }
    float* D = dst + ((size_t)y * cols + x) * 5;
    D[1] = (float)(b2 * q.ig11);
    D[0] = (float)(b3 * q.ig11);
    D[3] = (float)(b1 * q.ig03 + b4 * q.ig33);
    D[2] = (float)(b1 * q.ig03 + b5 * q.ig33);
    D[4] = (float)(b6 * q.ig55);
}

// displaced bilinear sample of the second expansion, averaged with the first; the matrix entries
// G11, G12, G22, h1, h2 per pixel
__global__ void k_fb_update_matrices(const float* __restrict__ R0, const float* __restrict__ R1,
                                     const float* __restrict__ flow, int rows, int cols, float* __restrict__ M)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows)
        return;
    const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    const size_t step1 = (size_t)cols * 5;
    const float* r0 = R0 + ((size_t)y * cols + x) * 5;
    const float2 d = ((const float2*)flow)[(size_t)y * cols + x];
    const float dx = d.x, dy = d.y;
    float fx = x + dx, fy = y + dy;
    const double flx = floor((double)fx), fly = floor((double)fy);
    const int x1 = (flx >= -1e9 && flx <= 1e9) ? (int)flx : -1;
    const int y1 = (fly >= -1e9 && fly <= 1e9) ? (int)fly : -1;
    float r2, r3, r4, r5, r6;
    fx -= x1;
    fy -= y1;
    if ((unsigned)x1 < (unsigned)(cols - 1) && (unsigned)y1 < (unsigned)(rows - 1))
    {
        const float* p = R1 + (size_t)y1 * step1 + (size_t)x1 * 5;
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        r2 = a00 * p[0] + a01 * p[5] + a10 * p[step1] + a11 * p[step1 + 5];
        r3 = a00 * p[1] + a01 * p[6] + a10 * p[step1 + 1] + a11 * p[step1 + 6];
        r4 = a00 * p[2] + a01 * p[7] + a10 * p[step1 + 2] + a11 * p[step1 + 7];
        r5 = a00 * p[3] + a01 * p[8] + a10 * p[step1 + 3] + a11 * p[step1 + 8];
        r6 = a00 * p[4] + a01 * p[9] + a10 * p[step1 + 4] + a11 * p[step1 + 9];
        r4 = (r0[2] + r4) * 0.5f;
        r5 = (r0[3] + r5) * 0.5f;
        r6 = (r0[4] + r6) * 0.25f;
    }
    else
    {
        r2 = r3 = 0.f;
        r4 = r0[2];
        r5 = r0[3];
        r6 = r0[4] * 0.5f;
    }
    r2 = (r0[0] - r2) * 0.5f;
    r3 = (r0[1] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    if ((unsigned)(x - 5) >= (unsigned)(cols - 10) || (unsigned)(y - 5) >= (unsigned)(rows - 10))
    {
        const float scale = (x < 5 ? border[x] : 1.f) * (x >= cols - 5 ? border[cols - x - 1] : 1.f) *
                            (y < 5 ? border[y] : 1.f) * (y >= rows - 5 ? border[rows - y - 1] : 1.f);
        r2 *= scale, r3 *= scale, r4 *= scale, r5 *= scale, r6 *= scale;
    }
    float* m = M + ((size_t)y * cols + x) * 5;
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}

// box sums of the matrix entries down the window's rows (double, top to bottom, replicated rows);
// one thread per entry
__global__ void k_fb_box_rows(const float* __restrict__ M, int rows, int cols5, int m, double* __restrict__ vs)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols5 || y >= rows)
        return;
    double s = 0;
    for (int d = -m; d <= m; ++d)
        s += M[(size_t)clampi(y + d, 0, rows - 1) * cols5 + x];
    vs[(size_t)y * cols5 + x] = s;
}

// sums along the window's columns (left to right, replicated columns) and the 2x2 solve
__global__ void k_fb_box_solve(const double* __restrict__ vs, int rows, int cols, int m, double scale,
                               float* __restrict__ flow)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows)
        return;
    const double* v = vs + (size_t)y * cols * 5;
    double g[5] = {0, 0, 0, 0, 0};
    for (int d = -m; d <= m; ++d)
    {
        const double* c = v + (size_t)clampi(x + d, 0, cols - 1) * 5;
#pragma unroll
        for (int e = 0; e < 5; ++e)
            g[e] += c[e];
    }
    const double g11 = g[0] * scale, g12 = g[1] * scale, g22 = g[2] * scale, h1 = g[3] * scale, h2 = g[4] * scale;
    const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
    ((float2*)flow)[(size_t)y * cols + x] =
        make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
}

// Gaussian window (OPTFLOW_FARNEBACK_GAUSSIAN): half kernel of the window, float
struct FbWindow
{
    float k[128]; // winSize <= 255
    int m;
};

// window pass down the rows: centre tap plus symmetric pairs, replicated rows; one thread per entry
__global__ void k_fb_gauss_rows(const float* __restrict__ M, int rows, int cols5, FbWindow w, float* __restrict__ vs)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols5 || y >= rows)
        return;
    float s0 = M[(size_t)y * cols5 + x] * w.k[0];
    for (int i = 1; i <= w.m; ++i)
        s0 += (M[(size_t)clampi(y + i, 0, rows - 1) * cols5 + x] + M[(size_t)clampi(y - i, 0, rows - 1) * cols5 + x]) *
              w.k[i];
    vs[(size_t)y * cols5 + x] = s0;
}

// window pass along the columns and the 2x2 solve
__global__ void k_fb_gauss_solve(const float* __restrict__ vs, int rows, int cols, FbWindow w, float* __restrict__ flow)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows)
        return;
    const float* v = vs + (size_t)y * cols * 5;
    float h[5];
#pragma unroll
    for (int e = 0; e < 5; ++e)
    {
        float sum = v[x * 5 + e] * w.k[0];
        for (int i = 1; i <= w.m; ++i)
            sum += w.k[i] * (v[clampi(x - i, 0, cols - 1) * 5 + e] + v[clampi(x + i, 0, cols - 1) * 5 + e]);
        h[e] = sum;
    }
    const double g11 = h[0], g12 = h[1], g22 = h[2], h1 = h[3], h2 = h[4];
    const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
    ((float2*)flow)[(size_t)y * cols + x] =
        make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
}

void window_setup(int win, FbWindow& w)
{
    const int m = win / 2;
    const double sigma = m * 0.3;
    double s = 1;
    w.k[0] = (float)s;
    for (int i = 1; i <= m; ++i)
    {
        const float t = (float)std::exp(-i * i / (2 * sigma * sigma));
        w.k[i] = t;
        s += t * 2;
    }
    s = 1. / s;
    for (int i = 0; i <= m; ++i)
        w.k[i] = (float)(w.k[i] * s);
    w.m = m;
}

constexpr int kFlagGaussian = 256; // OPTFLOW_FARNEBACK_GAUSSIAN

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

int check_params(const FarnebackParams& p)
{
    if (!(p.pyrScale > 0 && p.pyrScale < 1))
        return fb_bad("Farneback: pyrScale must lie in (0, 1)");
    if (p.numLevels < 0 || p.winSize < 1 || p.winSize > 255 || p.numIters < 0)
        return fb_bad("Farneback: numLevels >= 0, 1 <= winSize <= 255, numIters >= 0 required");
    if (p.polyN < 1 || p.polyN > kMaxPoly)
        return fb_bad("Farneback: polyN must lie in 1..16");
    if (p.flags != 0 && p.flags != kFlagGaussian)
        return fb_bad("Farneback: flags must be 0 (box window) or 256 (Gaussian window); an initial flow is not built");
    return SRX_OK;
}

} // namespace

int farneback_set_int(FarnebackParams& p, const std::string& n, int v, bool* known)
{
    *known = true;
    FarnebackParams q = p;
    if (n == "numLevels")
        q.numLevels = v;
    else if (n == "winSize")
        q.winSize = v;
    else if (n == "numIters")
        q.numIters = v;
    else if (n == "polyN")
        q.polyN = v;
    else if (n == "flags")
        q.flags = v;
    else
    {
        *known = false;
        return SRX_OK;
    }
    if (int rc = check_params(q))
        return rc;
    p = q;
    return SRX_OK;
}

int farneback_set_double(FarnebackParams& p, const std::string& n, double v, bool* known)
{
    *known = true;
    FarnebackParams q = p;
    if (n == "pyrScale")
        q.pyrScale = v;
    else if (n == "polySigma")
        q.polySigma = v;
    else
    {
        *known = false;
        return SRX_OK;
    }
    if (n == "polySigma" && !(v >= 0))
        return fb_bad("Farneback: polySigma must not be negative");
    if (int rc = check_params(q))
        return rc;
    p = q;
    return SRX_OK;
}

bool farneback_get_int(const FarnebackParams& p, const std::string& n, int* v)
{
    if (n == "numLevels")
        *v = p.numLevels;
    else if (n == "winSize")
        *v = p.winSize;
    else if (n == "numIters")
        *v = p.numIters;
    else if (n == "polyN")
        *v = p.polyN;
    else if (n == "flags")
        *v = p.flags;
    else
        return false;
    return true;
}

bool farneback_get_double(const FarnebackParams& p, const std::string& n, double* v)
{
    if (n == "pyrScale")
        *v = p.pyrScale;
    else if (n == "polySigma")
        *v = p.polySigma;
    else
        return false;
    return true;
}

struct Farneback::Impl
{
    DeviceBuffer gray[2], raw[2], tmp, blurred, level, rowbuf, R[2], M, vs, flow[2];
};

void Farneback::release()
{
    impl_.reset();
}

int Farneback::calc_images(const srx_image& f0, const srx_image& f1, const FarnebackParams& p, float* flow_uv,
                           hipStream_t st)
{
    if (!impl_)
        impl_ = std::make_shared<Impl>();
    Impl& m = *impl_;
    const int rows = f0.rows, cols = f0.cols, ch = f0.channels;
    if (ch != 1 && ch != 3 && ch != 4)
        return fb_bad("flow input must have 1, 3 or 4 channels");
    const size_t px = (size_t)rows * cols;
    const size_t esz = f0.depth == SRX_8U ? 1 : (f0.depth == SRX_16U ? 2 : 4);
    const size_t rowbytes = (size_t)cols * ch * esz;
    const srx_image* fr[2] = {&f0, &f1};
    for (int i = 0; i < 2; ++i)
    {
        if (int rc = m.gray[i].reserve(px * sizeof(float)))
            return rc;
        if (int rc = m.raw[i].reserve(rowbytes * rows))
            return rc;
        const size_t step = fr[i]->step ? fr[i]->step : rowbytes;
        SRX_HIP(copy_pitched(m.raw[i].get(), rowbytes, fr[i]->data, step, rowbytes, rows,
                             fr[i]->mem == SRX_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
        if (int rc = launch_to_gray(m.raw[i].get(), px, ch, f0.depth, m.gray[i].as<float>(), st))
            return rc;
    }
    return calc_gray(nullptr, nullptr, rows, cols, p, flow_uv, st);
}

// I0 / I1 == nullptr: the grey images are already in place (calc_images)
int Farneback::calc_gray(const float* I0, const float* I1, int rows, int cols, const FarnebackParams& p,
                         float* flow_uv, hipStream_t st)
{
    if (!impl_)
        impl_ = std::make_shared<Impl>();
    Impl& m = *impl_;
    if (rows < 1 || cols < 1 || !flow_uv)
        return fb_bad("Farneback: bad arguments");
    if (int rc = check_params(p))
        return rc;
    const size_t px0 = (size_t)rows * cols;
    if (I0)
        for (int i = 0; i < 2; ++i)
        {
            if (int rc = m.gray[i].reserve(px0 * sizeof(float)))
                return rc;
            SRX_HIP(hipMemcpyAsync(m.gray[i].get(), i ? I1 : I0, px0 * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
    for (DeviceBuffer* d : {&m.tmp, &m.blurred, &m.level})
        if (int rc = d->reserve(px0 * sizeof(float)))
            return rc;
    if (int rc = m.rowbuf.reserve(px0 * 3 * sizeof(float)))
        return rc;
    for (DeviceBuffer* d : {&m.R[0], &m.R[1], &m.M})
        if (int rc = d->reserve(px0 * 5 * sizeof(float)))
            return rc;
    if (int rc = m.vs.reserve(px0 * 5 * sizeof(double)))
        return rc;
    for (int i = 0; i < 2; ++i)
        if (int rc = m.flow[i].reserve(px0 * 2 * sizeof(float)))
            return rc;

    // levels below the full size that are used (calcOpticalFlowFarneback's min_size rule)
    int levels = 0;
    {
        double scale = 1;
        for (; levels < p.numLevels; ++levels)
        {
            scale *= p.pyrScale;
            if (cols * scale < 32 || rows * scale < 32)
                break;
        }
    }
    FbPoly q;
    poly_setup(p.polyN, p.polySigma, q);
    const dim3 b(64, 4);
    int cur = 0, prows = 0, pcols = 0;
    for (int k = levels; k >= 0; --k)
    {
        double scale = 1;
        for (int i = 0; i < k; ++i)
            scale *= p.pyrScale;
        FbTaps taps;
        if (!smooth_taps(scale, taps))
            return fb_bad("Farneback: the blur ahead of the coarsest level needs more than 127 taps");
        const int width = (int)std::nearbyint(cols * scale), height = (int)std::nearbyint(rows * scale);
        const dim3 g = grid2(width, height, b), g0 = grid2(cols, rows, b);
        float* flow = k == 0 ? flow_uv : m.flow[cur].as<float>();
        if (!prows)
            SRX_HIP(hipMemsetAsync(flow, 0, (size_t)width * height * 2 * sizeof(float), st));
        else
        {
            hipLaunchKernelGGL(k_fb_resize_linear<2>, g, b, 0, st, m.flow[cur ^ 1].as<float>(), prows, pcols, flow,
                               height, width, 1.0 / ((double)width / pcols), 1.0 / ((double)height / prows),
                               1. / p.pyrScale);
            SRX_LAUNCH_CHECK("farneback flow resize");
        }
        for (int i = 0; i < 2; ++i)
        {
            hipLaunchKernelGGL(k_fb_blur_rows, g0, b, 0, st, m.gray[i].as<float>(), rows, cols, taps,
                               m.tmp.as<float>());
            hipLaunchKernelGGL(k_fb_blur_cols, g0, b, 0, st, m.tmp.as<float>(), rows, cols, taps,
                               m.blurred.as<float>());
            hipLaunchKernelGGL(k_fb_resize_linear<1>, g, b, 0, st, m.blurred.as<float>(), rows, cols,
                               m.level.as<float>(), height, width, 1.0 / ((double)width / cols),
                               1.0 / ((double)height / rows), 0.0);
            hipLaunchKernelGGL(k_fb_poly_rows, g, b, 0, st, m.level.as<float>(), height, width, q,
                               m.rowbuf.as<float>());
            hipLaunchKernelGGL(k_fb_poly_cols, g, b, 0, st, m.rowbuf.as<float>(), height, width, q,
                               m.R[i].as<float>());
            SRX_LAUNCH_CHECK("farneback expansion");
        }
        hipLaunchKernelGGL(k_fb_update_matrices, g, b, 0, st, m.R[0].as<float>(), m.R[1].as<float>(), flow, height,
                           width, m.M.as<float>());
        const int half = p.winSize / 2;
        const double norm = 1. / ((double)p.winSize * p.winSize);
        FbWindow win;
        if (p.flags & kFlagGaussian)
            window_setup(p.winSize, win);
        for (int i = 0; i < p.numIters; ++i)
        {
            if (p.flags & kFlagGaussian)
            {
                hipLaunchKernelGGL(k_fb_gauss_rows, grid2(width * 5, height, b), b, 0, st, m.M.as<float>(), height,
                                   width * 5, win, m.vs.as<float>());
                hipLaunchKernelGGL(k_fb_gauss_solve, g, b, 0, st, m.vs.as<float>(), height, width, win, flow);
            }
            else
            {
                hipLaunchKernelGGL(k_fb_box_rows, grid2(width * 5, height, b), b, 0, st, m.M.as<float>(), height,
                                   width * 5, half, m.vs.as<double>());
                hipLaunchKernelGGL(k_fb_box_solve, g, b, 0, st, m.vs.as<double>(), height, width, half, norm, flow);
            }
            if (i < p.numIters - 1)
                hipLaunchKernelGGL(k_fb_update_matrices, g, b, 0, st, m.R[0].as<float>(), m.R[1].as<float>(), flow,
                                   height, width, m.M.as<float>());
        }
        SRX_LAUNCH_CHECK("farneback update");
        prows = height, pcols = width;
        cur ^= 1;
    }
    return SRX_OK;
}

} // namespace srx
