/*
 * CPU oracle for the built-in Farneback dense optical flow (polynomial expansion,
 * G. Farneback, "Two-Frame Motion Estimation Based on Polynomial Expansion", SCIA 2003).
 *
 * TEST INFRASTRUCTURE ONLY (see btvl1_oracle.c).
 *
 * PARITY UNPINNED.  The reference's default flow is its class Farneback
 * (lib/src/btv_l1.cpp:287), a wrapper around OpenCV's calcOpticalFlowFarneback
 * (lib/src/optical_flow.cpp:41-50,84-87: pyrScale 0.5, numLevels 5, winSize 13, numIters 10,
 * polyN 5, polySigma 1.1, flags 0).  The arithmetic lives in OpenCV (version unpinned, not in
 * the tree, not installed here) and the reference holds no expected flow fields.  This file
 * restates that routine from the published method and the structure of OpenCV 2.4's
 * implementation as far as it is known here; the GPU implementation is checked against this
 * file bit for bit and against nothing else.
 *
 * Frozen discretisation
 *   input      8-bit grey as convertToType(frame, CV_8U, 1) gives it (orc_tvl1_to_gray), as float.
 *   levels     k = numLevels..0 after dropping levels whose scaled width or height is below 32;
 *              scale = pyrScale^k; level size = round-half-even(size * scale).
 *   level image  Gaussian blur of the FULL-SIZE image with sigma = (1/scale - 1)/2 and size
 *              max(round(5 sigma) | 1, 3) (getGaussianKernel in float: fixed {1,2,1}/4 at sigma 0;
 *              row pass then column pass, BORDER_REFLECT_101; row taps summed symmetrically for
 *              sizes <= 5 and left to right for larger ones, column taps symmetrically), then
 *              bilinear resize (cv::resize INTER_LINEAR rules) to the level size.
 *   flow init  zero on the coarsest level, otherwise bilinear resize of the coarser flow times
 *              1/pyrScale (product in double, rounded to float).
 *   expansion  separable polynomial expansion with the (2 polyN + 1)-tap Gaussian of polySigma,
 *              replicated borders; vertical pass in float, horizontal pass accumulated in double;
 *              five coefficients per pixel (y, x, yy, xx, xy) through the inverse of the 6x6 moment
 *              matrix (here: closed form of its 3x3 block; OpenCV uses a Cholesky inverse).
 *   matrices   displaced bilinear sample of the second expansion, averaged with the first,
 *              border attenuation {0.14, 0.14, 0.4472, 0.4472, 0.4472}; all float.
 *   update     numIters times: box mean of the five matrix entries over winSize x winSize with
 *              replicated borders, summed in double (rows top to bottom, then columns left to
 *              right; OpenCV keeps running sums, which can differ in the last bit of a double),
 *              2x2 solve with 1e-3 added to the determinant; matrices recomputed from the new
 *              flow after every iteration but the last.  With flags = 256
 *              (OPTFLOW_FARNEBACK_GAUSSIAN) the window is a winSize-tap Gaussian of sigma
 *              0.3 (winSize / 2) instead of the box: float, vertical pass then horizontal pass, centre tap plus
 *              symmetric pairs, replicated borders.  Flag 4 (initial flow) is not restated.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

int orc_border_reflect101(int p, int len);
void orc_resize_linear(const float* src, int srows, int scols, float* dst, int drows, int dcols);

typedef struct orc_farneback_params
{
    double pyrScale;
    int numLevels, winSize, numIters, polyN;
    double polySigma;
    int flags;
} orc_farneback_params;

#define FB_MAX_TAPS 255
#define FB_MAX_POLY 16

static int clampi(int v, int lo, int hi)
{
    return v < lo ? lo : v > hi ? hi : v;
}

static int cv_round(double v)
{
    return (int)nearbyint(v);
}

/* getGaussianKernel(n, sigma, CV_32F) */
static void gaussian_taps(int n, double sigma, float* k)
{
    static const float k3[3] = {0.25f, 0.5f, 0.25f};
    const int fixed = n == 3 && sigma <= 0;
    const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; ++i)
    {
        const double x = i - (n - 1) * 0.5;
        const double t = fixed ? (double)k3[i] : exp(scale2X * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i)
        k[i] = (float)(k[i] * sum);
}

/* size and taps of the blur that precedes the resize to the level of the given scale */
ORC_API int orc_farneback_smooth_taps(double scale, float* k)
{
    const double sigma = (1. / scale - 1) * 0.5;
    int n = cv_round(sigma * 5) | 1;
    if (n < 3)
        n = 3;
    if (n > FB_MAX_TAPS)
        return -1;
    if (k)
        gaussian_taps(n, sigma, k);
    return n;
}

ORC_API void orc_farneback_blur(const float* src, int rows, int cols, const float* k, int n, float* dst)
{
    const int h = n / 2;
    float* tmp = (float*)malloc(sizeof(float) * (size_t)rows * cols);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
    {
        const float* S = src + (size_t)y * cols;
        float* D = tmp + (size_t)y * cols;
        for (int x = 0; x < cols; ++x)
        {
            float s;
            if (n <= 5)
            {
                s = S[x] * k[h];
                for (int j = 1; j <= h; ++j)
                    s += (S[orc_border_reflect101(x - j, cols)] + S[orc_border_reflect101(x + j, cols)]) * k[h + j];
            }
            else
            {
                s = k[0] * S[orc_border_reflect101(x - h, cols)];
                for (int j = 1; j < n; ++j)
                    s += k[j] * S[orc_border_reflect101(x - h + j, cols)];
            }
            D[x] = s;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
    {
        float* D = dst + (size_t)y * cols;
        for (int x = 0; x < cols; ++x)
        {
            float s = tmp[(size_t)y * cols + x] * k[h];
            for (int j = 1; j <= h; ++j)
                s += (tmp[(size_t)orc_border_reflect101(y - j, rows) * cols + x] +
                      tmp[(size_t)orc_border_reflect101(y + j, rows) * cols + x]) *
                     k[h + j];
            D[x] = s;
        }
    }
    free(tmp);
}

/* Taps g, x*g, x*x*g (index 0..n, the kernel is symmetric) and the four entries of the inverse
 * moment matrix that the expansion uses. */
ORC_API int orc_farneback_poly_setup(int n, double sigma, float* g, float* xg, float* xxg, double* ig)
{
    if (n < 1 || n > FB_MAX_POLY)
        return -1;
    float gf[2 * FB_MAX_POLY + 1];
    if (sigma < FLT_EPSILON)
        sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; ++x)
    {
        gf[x + n] = (float)exp(-x * x / (2 * sigma * sigma));
        s += gf[x + n];
    }
    s = 1. / s;
    for (int x = -n; x <= n; ++x)
        gf[x + n] = (float)(gf[x + n] * s);
    for (int x = 0; x <= n; ++x)
    {
        g[x] = gf[x + n];
        xg[x] = (float)(x * gf[x + n]);
        xxg[x] = (float)(x * x * gf[x + n]);
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
    /* moment matrix over the basis (1, x, y, xx, yy, xy):
     *   rows/cols {0,3,4}: [G00 G11 G11; G11 G33 G55; G11 G55 G33], (1,1) = (2,2) = G11, (5,5) = G55.
     * Inverse of the 3x3 block, by symmetry of its last two rows: with a = G00, b = G11, c = G33,
     * d = G55:  inv(0,3) = -b / (a (c + d) - 2 b b),  inv(3,3) = (a c - b b) / ((c - d)(a (c + d) - 2 b b)). */
    const double a = G00, b = G11, c = G33, d = G55;
    const double den = a * (c + d) - 2 * b * b;
    ig[0] = 1. / G11;                         /* ig11 */
    ig[1] = -b / den;                         /* ig03 */
    ig[2] = (a * c - b * b) / ((c - d) * den); /* ig33 */
    ig[3] = 1. / G55;                         /* ig55 */
    return 0;
}

/* src: rows x cols; dst: rows x cols x 5 */
ORC_API void orc_farneback_poly_exp(const float* src, int rows, int cols, int n, const float* g, const float* xg,
                                    const float* xxg, const double* ig, float* dst)
{
    const double ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];
#pragma omp parallel
    {
        float* row = (float*)malloc(sizeof(float) * (size_t)cols * 3);
#pragma omp for schedule(static)
        for (int y = 0; y < rows; ++y)
        {
            const float* c0 = src + (size_t)y * cols;
            for (int x = 0; x < cols; ++x)
            {
                row[x * 3] = c0[x] * g[0];
                row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
            }
            for (int k = 1; k <= n; ++k)
            {
                const float* s0 = src + (size_t)clampi(y - k, 0, rows - 1) * cols;
                const float* s1 = src + (size_t)clampi(y + k, 0, rows - 1) * cols;
                for (int x = 0; x < cols; ++x)
                {
                    const float p = s0[x] + s1[x];
                    const float t0 = row[x * 3] + g[k] * p;
                    const float t1 = row[x * 3 + 1] + xg[k] * (s1[x] - s0[x]);
                    const float t2 = row[x * 3 + 2] + xxg[k] * p;
                    row[x * 3] = t0, row[x * 3 + 1] = t1, row[x * 3 + 2] = t2;
                }
            }
            float* D = dst + (size_t)y * cols * 5;
            for (int x = 0; x < cols; ++x)
            {
                double b1 = row[x * 3] * g[0], b2 = 0, b3 = row[x * 3 + 1] * g[0], b4 = 0, b5 = row[x * 3 + 2] * g[0],
                       b6 = 0;
                for (int k = 1; k <= n; ++k)
                {
                    const float* rp = row + clampi(x + k, 0, cols - 1) * 3;
                    const float* rm = row + clampi(x - k, 0, cols - 1) * 3;
                    const double tg = rp[0] + rm[0];
                    b1 += tg * g[k];
                    b4 += tg * xxg[k];
                    b2 += (rp[0] - rm[0]) * xg[k];
                    b3 += (rp[1] + rm[1]) * g[k];
                    b6 += (rp[1] - rm[1]) * xg[k];
                    b5 += (rp[2] + rm[2]) * g[k];
                }
                D[x * 5 + 1] = (float)(b2 * ig11);
                D[x * 5] = (float)(b3 * ig11);
                D[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
                D[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
                D[x * 5 + 4] = (float)(b6 * ig55);
            }
        }
        free(row);
    }
}

/* R0, R1, M: rows x cols x 5; flow: rows x cols x 2 */
ORC_API void orc_farneback_update_matrices(const float* R0, const float* R1, const float* flow, int rows, int cols,
                                           float* M)
{
    static const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    const size_t step1 = (size_t)cols * 5;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
    {
        const float* fl = flow + (size_t)y * cols * 2;
        const float* r0 = R0 + (size_t)y * cols * 5;
        float* m = M + (size_t)y * cols * 5;
        for (int x = 0; x < cols; ++x)
        {
            const float dx = fl[x * 2], dy = fl[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            /* floor of a float that may be huge or NaN: keep the comparison below meaningful */
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
                r4 = (r0[x * 5 + 2] + r4) * 0.5f;
                r5 = (r0[x * 5 + 3] + r5) * 0.5f;
                r6 = (r0[x * 5 + 4] + r6) * 0.25f;
            }
            else
            {
                r2 = r3 = 0.f;
                r4 = r0[x * 5 + 2];
                r5 = r0[x * 5 + 3];
                r6 = r0[x * 5 + 4] * 0.5f;
            }
            r2 = (r0[x * 5] - r2) * 0.5f;
            r3 = (r0[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - 5) >= (unsigned)(cols - 10) || (unsigned)(y - 5) >= (unsigned)(rows - 10))
            {
                const float scale = (x < 5 ? border[x] : 1.f) * (x >= cols - 5 ? border[cols - x - 1] : 1.f) *
                                    (y < 5 ? border[y] : 1.f) * (y >= rows - 5 ? border[rows - y - 1] : 1.f);
                r2 *= scale, r3 *= scale, r4 *= scale, r5 *= scale, r6 *= scale;
            }
            m[x * 5] = r4 * r4 + r6 * r6;
            m[x * 5 + 1] = (r4 + r5) * r6;
            m[x * 5 + 2] = r5 * r5 + r6 * r6;
            m[x * 5 + 3] = r4 * r2 + r6 * r3;
            m[x * 5 + 4] = r6 * r2 + r5 * r3;
        }
    }
}

/* One flow update from the box-averaged matrices. */
ORC_API void orc_farneback_update_flow(const float* M, int rows, int cols, int win, float* flow)
{
    const int m = win / 2;
    const double scale = 1. / ((double)win * win);
    double* vs = (double*)malloc(sizeof(double) * (size_t)rows * cols * 5);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols * 5; ++x)
        {
            double s = 0;
            for (int d = -m; d <= m; ++d)
                s += M[(size_t)clampi(y + d, 0, rows - 1) * cols * 5 + x];
            vs[(size_t)y * cols * 5 + x] = s;
        }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
    {
        const double* v = vs + (size_t)y * cols * 5;
        for (int x = 0; x < cols; ++x)
        {
            double g[5] = {0, 0, 0, 0, 0};
            for (int d = -m; d <= m; ++d)
            {
                const double* c = v + (size_t)clampi(x + d, 0, cols - 1) * 5;
                for (int e = 0; e < 5; ++e)
                    g[e] += c[e];
            }
            const double g11 = g[0] * scale, g12 = g[1] * scale, g22 = g[2] * scale, h1 = g[3] * scale,
                         h2 = g[4] * scale;
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            flow[((size_t)y * cols + x) * 2] = (float)((g11 * h2 - g12 * h1) * idet);
            flow[((size_t)y * cols + x) * 2 + 1] = (float)((g22 * h1 - g12 * h2) * idet);
        }
    }
    free(vs);
}

/* The same with a Gaussian window (OPTFLOW_FARNEBACK_GAUSSIAN). */
ORC_API void orc_farneback_update_flow_gaussian(const float* M, int rows, int cols, int win, float* flow)
{
    const int m = win / 2;
    const double sigma = m * 0.3;
    float* kernel = (float*)malloc(sizeof(float) * (size_t)(m + 1));
    double s = 1;
    kernel[0] = (float)s;
    for (int i = 1; i <= m; ++i)
    {
        const float t = (float)exp(-i * i / (2 * sigma * sigma));
        kernel[i] = t;
        s += t * 2;
    }
    s = 1. / s;
    for (int i = 0; i <= m; ++i)
        kernel[i] = (float)(kernel[i] * s);
    float* vs = (float*)malloc(sizeof(float) * (size_t)rows * cols * 5);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols * 5; ++x)
        {
            float s0 = M[(size_t)y * cols * 5 + x] * kernel[0];
            for (int i = 1; i <= m; ++i)
                s0 += (M[(size_t)clampi(y + i, 0, rows - 1) * cols * 5 + x] +
                       M[(size_t)clampi(y - i, 0, rows - 1) * cols * 5 + x]) *
                      kernel[i];
            vs[(size_t)y * cols * 5 + x] = s0;
        }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
    {
        const float* v = vs + (size_t)y * cols * 5;
        for (int x = 0; x < cols; ++x)
        {
            float h[5];
            for (int e = 0; e < 5; ++e)
            {
                float sum = v[x * 5 + e] * kernel[0];
                for (int i = 1; i <= m; ++i)
                    sum += kernel[i] * (v[clampi(x - i, 0, cols - 1) * 5 + e] + v[clampi(x + i, 0, cols - 1) * 5 + e]);
                h[e] = sum;
            }
            const double g11 = h[0], g12 = h[1], g22 = h[2], h1 = h[3], h2 = h[4];
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            flow[((size_t)y * cols + x) * 2] = (float)((g11 * h2 - g12 * h1) * idet);
            flow[((size_t)y * cols + x) * 2 + 1] = (float)((g22 * h1 - g12 * h2) * idet);
        }
    }
    free(vs), free(kernel);
}

/* number of pyramid levels below the full size that are used (calcOpticalFlowFarneback's
 * min_size rule) */
ORC_API int orc_farneback_levels(int rows, int cols, double pyr_scale, int levels)
{
    int k;
    double scale = 1;
    for (k = 0; k < levels; ++k)
    {
        scale *= pyr_scale;
        if (cols * scale < 32 || rows * scale < 32)
            break;
    }
    return k;
}

/* Flow I0 -> I1 of two grey float images; flow: rows x cols x 2 interleaved. */
ORC_API int orc_farneback_calc(const float* I0, const float* I1, int rows, int cols, const orc_farneback_params* p,
                               float* flow_out)
{
    if (rows < 1 || cols < 1 || !(p->pyrScale > 0 && p->pyrScale < 1) || p->winSize < 1 || p->numIters < 0 ||
        p->polyN < 1 || p->polyN > FB_MAX_POLY || (p->flags != 0 && p->flags != 256))
        return -1;
    const int levels = orc_farneback_levels(rows, cols, p->pyrScale, p->numLevels);
    float g[FB_MAX_POLY + 1], xg[FB_MAX_POLY + 1], xxg[FB_MAX_POLY + 1];
    double ig[4];
    if (orc_farneback_poly_setup(p->polyN, p->polySigma, g, xg, xxg, ig))
        return -1;
    const size_t px0 = (size_t)rows * cols;
    const float* img[2] = {I0, I1};
    float* blurred = (float*)malloc(sizeof(float) * px0);
    float* I = (float*)malloc(sizeof(float) * px0);
    float* R[2] = {(float*)malloc(sizeof(float) * px0 * 5), (float*)malloc(sizeof(float) * px0 * 5)};
    float* M = (float*)malloc(sizeof(float) * px0 * 5);
    float* flow = (float*)malloc(sizeof(float) * px0 * 2);
    float* prev = (float*)malloc(sizeof(float) * px0 * 2);
    float* plane = (float*)malloc(sizeof(float) * px0 * 2);
    int prows = 0, pcols = 0;
    float taps[FB_MAX_TAPS];
    for (int k = levels; k >= 0; --k)
    {
        double scale = 1;
        for (int i = 0; i < k; ++i)
            scale *= p->pyrScale;
        const int nt = orc_farneback_smooth_taps(scale, taps);
        if (nt < 0)
            return -1;
        const int width = cv_round(cols * scale), height = cv_round(rows * scale);
        const size_t px = (size_t)width * height;
        if (!prows)
            memset(flow, 0, sizeof(float) * px * 2);
        else
        {
            /* resize(prevFlow, flow, INTER_LINEAR); flow *= 1 / pyr_scale */
            const size_t ppx = (size_t)prows * pcols;
            const double mul = 1. / p->pyrScale;
            for (int c = 0; c < 2; ++c)
            {
                for (size_t i = 0; i < ppx; ++i)
                    plane[i] = prev[i * 2 + c];
                orc_resize_linear(plane, prows, pcols, plane + ppx, height, width);
                for (size_t i = 0; i < px; ++i)
                    flow[i * 2 + c] = (float)(plane[ppx + i] * mul);
            }
        }
        for (int i = 0; i < 2; ++i)
        {
            orc_farneback_blur(img[i], rows, cols, taps, nt, blurred);
            orc_resize_linear(blurred, rows, cols, I, height, width);
            orc_farneback_poly_exp(I, height, width, p->polyN, g, xg, xxg, ig, R[i]);
        }
        orc_farneback_update_matrices(R[0], R[1], flow, height, width, M);
        for (int i = 0; i < p->numIters; ++i)
        {
            if (p->flags & 256)
                orc_farneback_update_flow_gaussian(M, height, width, p->winSize, flow);
            else
                orc_farneback_update_flow(M, height, width, p->winSize, flow);
            if (i < p->numIters - 1)
                orc_farneback_update_matrices(R[0], R[1], flow, height, width, M);
        }
        memcpy(prev, flow, sizeof(float) * px * 2);
        prows = height, pcols = width;
    }
    memcpy(flow_out, flow, sizeof(float) * px0 * 2);
    free(blurred), free(I), free(R[0]), free(R[1]), free(M), free(flow), free(prev), free(plane);
    return 0;
}
