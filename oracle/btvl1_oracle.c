/*
 * CPU oracle for the BTV-L1 multiframe super-resolution solver.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is a plain-C restatement of the
 * reference CPU path (lib/src/btv_l1.cpp) and of the OpenCV 2.4 primitives it
 * calls.  It is used by tests/, by __graft_entry__.smoke() and by bench.py's
 * cpu_baseline leg as the checker and the timed CPU baseline; nothing in the
 * shipped GPU path links, loads or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests or golden vectors and OpenCV
 * is not available to build it, so the OpenCV semantics below are restated
 * from the OpenCV 2.4 sources as the written spec.  Each function says which
 * reference lines and which OpenCV routine it follows, so that a reader with
 * OpenCV at hand can audit it.
 *
 * Pass structure is deliberately the reference's: one full pass over memory
 * per remap / blur / decimate / sign / zero-insert / blur / remap / add, per
 * window frame and per iteration.  Build with -ffp-contract=off: every
 * multiply and add below is a separately rounded IEEE operation, in the order
 * written.
 *
 * Images are dense row-major float32, channels interleaved:
 *   pixel (y, x), channel c  ->  data[(y * cols + x) * ch + c]
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* scalar helpers                                                            */

/* OpenCV cvRound(double): SSE2 cvtsd2si, round-half-to-even; out-of-range and
 * NaN give INT_MIN ("integer indefinite"). */
static int cv_round(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0))
        return INT_MIN;
    double r = nearbyint(v); /* default rounding mode: to nearest even */
    if (r >= 2147483648.0)
        return INT_MIN;
    return (int)r;
}

static int cv_floor(double v)
{
    return (int)floor(v);
}

static short saturate_short(int v)
{
    return (short)(v < SHRT_MIN ? SHRT_MIN : v > SHRT_MAX ? SHRT_MAX : v);
}

static unsigned char saturate_uchar(int v)
{
    return (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
}

/* OpenCV borderInterpolate(p, len, BORDER_REFLECT_101). */
ORC_API int orc_border_reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len)
        return p;
    if (len == 1)
        return 0;
    do
    {
        if (p < 0)
            p = -p - 1 + 1;
        else
            p = len - 1 - (p - len) - 1;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* OpenCV borderInterpolate(p, len, BORDER_WRAP); this is all that the ring
 * helper at() of lib/src/ring_buffer.hpp:36-46 does. */
ORC_API int orc_border_wrap(int p, int len)
{
    if ((unsigned)p < (unsigned)len)
        return p;
    if (p < 0)
        p -= ((p - len + 1) / len) * len;
    if (p >= len)
        p %= len;
    return p;
}

ORC_API void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0)
        omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORC_API int orc_get_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* calcRelativeMotions: lib/src/btv_l1.cpp:36-57                             */
/* fwd: n-1 forward motions (frame i -> i+1), rel: n outputs, all rows*cols*2 */

ORC_API void orc_calc_relative_motions(const float* const* fwd, float* const* rel, int n, int base_idx, int rows,
                                       int cols)
{
    const size_t cnt = (size_t)rows * cols * 2;
    memset(rel[base_idx], 0, cnt * sizeof(float));
    for (int i = base_idx - 1; i >= 0; --i) /* add(rel[i+1], fwd[i], rel[i]) */
        for (size_t e = 0; e < cnt; ++e)
            rel[i][e] = rel[i + 1][e] + fwd[i][e];
    for (int i = base_idx + 1; i < n; ++i) /* subtract(rel[i-1], fwd[i-1], rel[i]) */
        for (size_t e = 0; e < cnt; ++e)
            rel[i][e] = rel[i - 1][e] - fwd[i - 1][e];
}

/* ------------------------------------------------------------------------ */
/* cv::resize(..., INTER_CUBIC) on float images, OpenCV 2.4 imgwarp.cpp:      */
/* interpolateCubic, the xofs/alpha set-up of cv::resize, HResizeCubic and    */
/* VResizeCubic.  Used at lib/src/btv_l1.cpp:75 and :412.                     */

static void cubic_coeffs(float x, float* c)
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

/* Horizontal tap table as cv::resize builds it: source index of tap 1 and the
 * four weights for every destination column.  OpenCV 2.4 snaps columns whose
 * tap 1 falls left of the image (sx < 0) or on/after the last column
 * (sx >= cols-1) to a plain copy (fx = 0); rows get no such treatment. */
ORC_API void orc_cubic_table_x(int scols, int dcols, int* sx_out, float* alpha_out)
{
    const double inv_scale = (double)dcols / scols;
    const double scale = 1. / inv_scale;
    for (int dx = 0; dx < dcols; ++dx)
    {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 1)
        {
            if (sx < 0)
                fx = 0, sx = 0;
        }
        if (sx + 2 >= scols)
        {
            if (sx >= scols - 1)
                fx = 0, sx = scols - 1;
        }
        sx_out[dx] = sx;
        cubic_coeffs(fx, alpha_out + 4 * dx);
    }
}

ORC_API void orc_cubic_table_y(int srows, int drows, int* sy_out, float* beta_out)
{
    const double inv_scale = (double)drows / srows;
    const double scale = 1. / inv_scale;
    for (int dy = 0; dy < drows; ++dy)
    {
        float fy = (float)((dy + 0.5) * scale - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        sy_out[dy] = sy;
        cubic_coeffs(fy, beta_out + 4 * dy);
    }
}

static int clampi(int v, int lo, int hi)
{
    return v < lo ? lo : v > hi ? hi : v;
}

ORC_API void orc_resize_cubic(const float* src, int srows, int scols, int ch, float* dst, int drows, int dcols)
{
    int* xofs = (int*)malloc(sizeof(int) * dcols);
    float* alpha = (float*)malloc(sizeof(float) * 4 * dcols);
    int* yofs = (int*)malloc(sizeof(int) * drows);
    float* beta = (float*)malloc(sizeof(float) * 4 * drows);
    float* hres = (float*)malloc(sizeof(float) * (size_t)srows * dcols * ch);
    orc_cubic_table_x(scols, dcols, xofs, alpha);
    orc_cubic_table_y(srows, drows, yofs, beta);

    /* HResizeCubic: every source row to destination width, float intermediate */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < srows; ++y)
    {
        const float* S = src + (size_t)y * scols * ch;
        float* D = hres + (size_t)y * dcols * ch;
        for (int dx = 0; dx < dcols; ++dx)
        {
            const float* a = alpha + 4 * dx;
            const int sx = xofs[dx];
            const int x0 = clampi(sx - 1, 0, scols - 1), x1 = clampi(sx, 0, scols - 1);
            const int x2 = clampi(sx + 1, 0, scols - 1), x3 = clampi(sx + 2, 0, scols - 1);
            for (int c = 0; c < ch; ++c)
                D[dx * ch + c] = S[x0 * ch + c] * a[0] + S[x1 * ch + c] * a[1] + S[x2 * ch + c] * a[2] +
                                 S[x3 * ch + c] * a[3];
        }
    }
    /* VResizeCubic: rows sy-1 .. sy+2 clipped to the image */
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < drows; ++dy)
    {
        const float* b = beta + 4 * dy;
        const int sy = yofs[dy];
        const float* S0 = hres + (size_t)clampi(sy - 1, 0, srows - 1) * dcols * ch;
        const float* S1 = hres + (size_t)clampi(sy, 0, srows - 1) * dcols * ch;
        const float* S2 = hres + (size_t)clampi(sy + 1, 0, srows - 1) * dcols * ch;
        const float* S3 = hres + (size_t)clampi(sy + 2, 0, srows - 1) * dcols * ch;
        float* D = dst + (size_t)dy * dcols * ch;
        for (int x = 0; x < dcols * ch; ++x)
            D[x] = S0[x] * b[0] + S1[x] * b[1] + S2[x] * b[2] + S3[x] * b[3];
    }
    free(xofs);
    free(alpha);
    free(yofs);
    free(beta);
    free(hres);
}

/* upscaleMotions: lib/src/btv_l1.cpp:59-78.  resize(INTER_CUBIC, fx=fy=scale)
 * then multiply by Scalar::all(scale) (float multiply). */
ORC_API void orc_upscale_motion(const float* lr, int rows, int cols, int scale, float* hr)
{
    const int R = rows * scale, C = cols * scale;
    orc_resize_cubic(lr, rows, cols, 2, hr, R, C);
    const float s = (float)scale;
    const size_t cnt = (size_t)R * C * 2;
    for (size_t e = 0; e < cnt; ++e)
        hr[e] = hr[e] * s;
}

/* buildMotionMaps: lib/src/btv_l1.cpp:80-101.  forward = (x,y) - m,
 * backward = (x,y) + m, both CV_32FC2. */
ORC_API void orc_build_motion_maps(const float* motion, int rows, int cols, float* fwd, float* bwd)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
        {
            const size_t e = ((size_t)y * cols + x) * 2;
            fwd[e] = (float)x - motion[e];
            fwd[e + 1] = (float)y - motion[e + 1];
            bwd[e] = (float)x + motion[e];
            bwd[e + 1] = (float)y + motion[e + 1];
        }
}

/* The integer source index that cv::remap(INTER_NEAREST) derives from a
 * CV_32FC2 map: convertTo(CV_16SC2) = saturate_cast<short>(cvRound(v)).
 * This is the "integer map construction" that has to match bit for bit. */
ORC_API void orc_map_to_short(const float* map, size_t count2, short* out)
{
    for (size_t e = 0; e < count2; ++e)
        out[e] = saturate_short(cv_round((double)map[e]));
}

/* cv::remap(src, dst, map, noArray(), INTER_NEAREST) with the default
 * BORDER_CONSTANT / Scalar() border, as called at lib/src/btv_l1.cpp:428,441.
 * dst has the map's size. */
ORC_API void orc_remap_nearest(const float* src, int srows, int scols, int ch, const float* map, int drows, int dcols,
                               float* dst)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x)
        {
            const size_t m = ((size_t)y * dcols + x) * 2;
            const int sx = saturate_short(cv_round((double)map[m]));
            const int sy = saturate_short(cv_round((double)map[m + 1]));
            float* D = dst + ((size_t)y * dcols + x) * ch;
            if ((unsigned)sx < (unsigned)scols && (unsigned)sy < (unsigned)srows)
            {
                const float* S = src + ((size_t)sy * scols + sx) * ch;
                for (int c = 0; c < ch; ++c)
                    D[c] = S[c];
            }
            else
                for (int c = 0; c < ch; ++c)
                    D[c] = 0.f;
        }
}

/* ------------------------------------------------------------------------ */
/* createGaussianFilter(type, Size(n,n), sigma) + FilterEngine::apply         */
/* (lib/src/btv_l1.cpp:394,430,439): getGaussianKernel in float, separable,   */
/* row pass then column pass, BORDER_REFLECT_101.                             */

ORC_API void orc_gaussian_kernel(int n, double sigma, float* k)
{
    static const float tab[4][7] = {{1.f},
                                    {0.25f, 0.5f, 0.25f},
                                    {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
                                    {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    const float* fixed = (n % 2 == 1 && n <= 7 && sigma <= 0) ? tab[n >> 1] : 0;
    const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; ++i)
    {
        const double x = i - (n - 1) * 0.5;
        const double t = fixed ? (double)fixed[i] : exp(scale2X * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i)
        k[i] = (float)(k[i] * sum);
}

/* One filtered sample of a 1-D symmetric kernel, in OpenCV's operation order.
 * Row filter, n <= 5  (SymmRowSmallFilter):   c*k0 + (l1+r1)*k1 + (l2+r2)*k2
 * Row filter, n >= 7  (RowFilter):            sequential, left tap first
 * Column filter, any n (SymmColumn[Small]Filter): c*k0 + (l1+r1)*k1 + ...    */
static float filt_sym(const float* v, const float* k, int n) /* v[0..n-1] taps, k full kernel */
{
    const int h = n / 2;
    float s = v[h] * k[h];
    for (int j = 1; j <= h; ++j)
        s += (v[h - j] + v[h + j]) * k[h + j];
    return s;
}

static float filt_seq(const float* v, const float* k, int n)
{
    float s = k[0] * v[0];
    for (int j = 1; j < n; ++j)
        s += k[j] * v[j];
    return s;
}

ORC_API void orc_sep_filter(const float* src, int rows, int cols, int ch, const float* k, int n, float* tmp,
                            float* dst)
{
    const int h = n / 2;
    /* row pass */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
    {
        const float* S = src + (size_t)y * cols * ch;
        float* D = tmp + (size_t)y * cols * ch;
        float v[32];
        for (int x = 0; x < cols; ++x)
            for (int c = 0; c < ch; ++c)
            {
                for (int j = 0; j < n; ++j)
                    v[j] = S[orc_border_reflect101(x - h + j, cols) * ch + c];
                D[x * ch + c] = n <= 5 ? filt_sym(v, k, n) : filt_seq(v, k, n);
            }
    }
    /* column pass */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
    {
        float* D = dst + (size_t)y * cols * ch;
        const float* R[32];
        for (int j = 0; j < n; ++j)
            R[j] = tmp + (size_t)orc_border_reflect101(y - h + j, rows) * cols * ch;
        for (int x = 0; x < cols * ch; ++x)
        {
            float s = R[h][x] * k[h];
            for (int j = 1; j <= h; ++j)
                s += (R[h - j][x] + R[h + j][x]) * k[h + j];
            D[x] = s;
        }
    }
}

/* cv::resize(src, dst, dsize, 0, 0, INTER_NEAREST), lib/src/btv_l1.cpp:432:
 * sx = min(floor(x * (1/inv_scale_x)), cols-1), same for rows. */
ORC_API void orc_resize_nearest(const float* src, int srows, int scols, int ch, float* dst, int drows, int dcols)
{
    const double ifx = 1. / ((double)dcols / scols), ify = 1. / ((double)drows / srows);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y)
    {
        int sy = cv_floor(y * ify);
        if (sy > srows - 1)
            sy = srows - 1;
        for (int x = 0; x < dcols; ++x)
        {
            int sx = cv_floor(x * ifx);
            if (sx > scols - 1)
                sx = scols - 1;
            for (int c = 0; c < ch; ++c)
                dst[((size_t)y * dcols + x) * ch + c] = src[((size_t)sy * scols + sx) * ch + c];
        }
    }
}

/* diffSign: lib/src/btv_l1.cpp:138-170.  a>b ? 1 : a<b ? -1 : 0 (NaN -> 0) */
ORC_API void orc_diff_sign(const float* a, const float* b, size_t count, float* dst)
{
    for (size_t e = 0; e < count; ++e)
        dst[e] = a[e] > b[e] ? 1.0f : a[e] < b[e] ? -1.0f : 0.0f;
}

/* upscale / upscaleImpl: lib/src/btv_l1.cpp:103-136.  Zero image, then
 * dst(y*scale, x*scale) = src(y, x). */
ORC_API void orc_upscale(const float* src, int rows, int cols, int ch, int scale, float* dst)
{
    const int C = cols * scale;
    memset(dst, 0, sizeof(float) * (size_t)rows * scale * C * ch);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
            for (int c = 0; c < ch; ++c)
                dst[((size_t)y * scale * C + (size_t)x * scale) * ch + c] = src[((size_t)y * cols + x) * ch + c];
}

/* calcBtvWeights: lib/src/btv_l1.cpp:172-189.  Enumeration m = 0..p,
 * l = p down to -m; returns the number of entries written (22 for size 7).
 * pow(float, int) is evaluated as in C++11: pow((double)alpha_f, (double)n)
 * rounded to float. */
ORC_API int orc_calc_btv_weights(int btv_kernel_size, double alpha, float* w)
{
    const int ksize = (btv_kernel_size - 1) / 2;
    const float alpha_f = (float)alpha;
    int ind = 0;
    for (int m = 0; m <= ksize; ++m)
        for (int l = ksize; l + m >= 0; --l, ++ind)
            w[ind] = (float)pow((double)alpha_f, (double)(abs(m) + abs(l)));
    return ind;
}

static float dsign(float a, float b)
{
    return a > b ? 1.0f : a < b ? -1.0f : 0.0f;
}

/* calcBtvRegularization / BtvRegularizationBody: lib/src/btv_l1.cpp:191-274.
 * dst is zeroed; interior pixels accumulate in enumeration order. */
ORC_API void orc_btv_regularization(const float* src, int rows, int cols, int ch, int btv_kernel_size, const float* w,
                                    float* dst)
{
    const int ksize = (btv_kernel_size - 1) / 2;
    memset(dst, 0, sizeof(float) * (size_t)rows * cols * ch);
#pragma omp parallel for schedule(static)
    for (int i = ksize; i < rows - ksize; ++i)
        for (int j = ksize; j < cols - ksize; ++j)
            for (int c = 0; c < ch; ++c)
            {
                const float v = src[((size_t)i * cols + j) * ch + c];
                float acc = dst[((size_t)i * cols + j) * ch + c];
                int ind = 0;
                for (int m = 0; m <= ksize; ++m)
                    for (int l = ksize; l + m >= 0; --l, ++ind)
                    {
                        const float p3 = src[((size_t)(i + m) * cols + (j + l)) * ch + c];
                        const float p2 = src[((size_t)(i - m) * cols + (j - l)) * ch + c];
                        acc += w[ind] * (dsign(v, p3) - dsign(p2, v));
                    }
                dst[((size_t)i * cols + j) * ch + c] = acc;
            }
}

/* cv::add on float images (lib/src/btv_l1.cpp:443) */
ORC_API void orc_add(const float* a, const float* b, size_t count, float* dst)
{
    for (size_t e = 0; e < count; ++e)
        dst[e] = a[e] + b[e];
}

/* cv::addWeighted on float images (lib/src/btv_l1.cpp:449,452):
 * dst = (float)(a*alpha + b*beta + gamma) evaluated in double. */
ORC_API void orc_add_weighted(const float* a, double alpha, const float* b, double beta, double gamma, size_t count,
                              float* dst)
{
#pragma omp parallel for schedule(static)
    for (long long e = 0; e < (long long)count; ++e)
        dst[e] = (float)((double)a[e] * alpha + (double)b[e] * beta + gamma);
}

/* Mat::convertTo(CV_8U) of a float image: cvRound (half to even), clamp. */
ORC_API void orc_convert_8u(const float* src, size_t count, unsigned char* dst)
{
    for (size_t e = 0; e < count; ++e)
        dst[e] = saturate_uchar(cv_round((double)src[e]));
}

/* Mat::convertTo(CV_32F) of an 8-bit image. */
ORC_API void orc_convert_32f(const unsigned char* src, size_t count, float* dst)
{
    for (size_t e = 0; e < count; ++e)
        dst[e] = (float)src[e];
}

/* ------------------------------------------------------------------------ */
/* BTV_L1_Base::run: lib/src/btv_l1.cpp:357-457                               */
/*                                                                            */
/* y[k]:   K low-resolution float frames, rows x cols x ch                    */
/* rel[k]: K relative motions (frame k -> base), rows x cols x 2              */
/* out:    (rows*scale - 2B) x (cols*scale - 2B) x ch float (cropped X)       */
/* x_full: optional (may be NULL) uncropped X after the last iteration        */

typedef struct orc_btv_params
{
    int scale;
    int iterations;
    double tau;
    double lambda;
    double alpha;
    int btv_kernel_size;
    int blur_kernel_size;
    double blur_sigma;
} orc_btv_params;

ORC_API int orc_btvl1_run(const float* const* y, const float* const* rel, int K, int base_idx, int rows, int cols,
                          int ch, const orc_btv_params* p, float* out, float* x_full)
{
    const int s = p->scale;
    const int R = rows * s, C = cols * s;
    const size_t hr = (size_t)R * C * ch, lr = (size_t)rows * cols * ch, hr2 = (size_t)R * C * 2;
    const int B = p->btv_kernel_size;
    if (K < 1 || base_idx < 0 || base_idx >= K || s < 2 || p->iterations < 1 || p->blur_kernel_size < 1 ||
        p->blur_kernel_size > 31 || !(p->blur_kernel_size & 1) || B < 1 || R - 2 * B < 1 || C - 2 * B < 1)
        return -1;

    /* upscaleMotions + buildMotionMaps (:383-388) */
    float** fwd = (float**)malloc(sizeof(float*) * K);
    float** bwd = (float**)malloc(sizeof(float*) * K);
    float* hrm = (float*)malloc(sizeof(float) * hr2);
    for (int k = 0; k < K; ++k)
    {
        fwd[k] = (float*)malloc(sizeof(float) * hr2);
        bwd[k] = (float*)malloc(sizeof(float) * hr2);
        orc_upscale_motion(rel[k], rows, cols, s, hrm);
        orc_build_motion_maps(hrm, R, C, fwd[k], bwd[k]);
    }
    free(hrm);

    /* blur filter and btv weights (:392-405) */
    float gk[32];
    orc_gaussian_kernel(p->blur_kernel_size, p->blur_sigma, gk);
    float* w = (float*)calloc((size_t)B * B + 1, sizeof(float));
    orc_calc_btv_weights(B, p->alpha, w);

    /* initial estimation (:412) */
    float* X = (float*)malloc(sizeof(float) * hr);
    orc_resize_cubic(y[base_idx], rows, cols, ch, X, R, C);

    float* diff = (float*)malloc(sizeof(float) * hr);
    float* a = (float*)malloc(sizeof(float) * hr);
    float* b = (float*)malloc(sizeof(float) * hr);
    float* tmp = (float*)malloc(sizeof(float) * hr);
    float* reg = (float*)malloc(sizeof(float) * hr);
    float* c = (float*)malloc(sizeof(float) * lr);

    for (int it = 0; it < p->iterations; ++it) /* :421-453 */
    {
        memset(diff, 0, sizeof(float) * hr);
        for (int k = 0; k < K; ++k)
        {
            orc_remap_nearest(X, R, C, ch, bwd[k], R, C, a);            /* a = M * Ih        */
            orc_sep_filter(a, R, C, ch, gk, p->blur_kernel_size, tmp, b); /* b = HM * Ih       */
            orc_resize_nearest(b, R, C, ch, c, rows, cols);             /* c = DHM * Ih      */
            orc_diff_sign(y[k], c, lr, c);                              /* sign(y - c)       */
            orc_upscale(c, rows, cols, ch, s, a);                       /* a = Dt * diff     */
            orc_sep_filter(a, R, C, ch, gk, p->blur_kernel_size, tmp, b); /* b = HtDt * diff   */
            orc_remap_nearest(b, R, C, ch, fwd[k], R, C, a);            /* a = MtHtDt * diff */
            orc_add(diff, a, hr, diff);
        }
        if (p->lambda > 0)
        {
            orc_btv_regularization(X, R, C, ch, B, w, reg);
            orc_add_weighted(diff, 1.0, reg, -p->lambda, 0.0, hr, diff);
        }
        orc_add_weighted(X, 1.0, diff, p->tau, 0.0, hr, X);
    }

    /* Rect inner(B, B, cols - 2B, rows - 2B) (:455-456) */
    const int oc = C - 2 * B, orr = R - 2 * B;
    for (int yy = 0; yy < orr; ++yy)
        memcpy(out + (size_t)yy * oc * ch, X + ((size_t)(yy + B) * C + B) * ch, sizeof(float) * (size_t)oc * ch);
    if (x_full)
        memcpy(x_full, X, sizeof(float) * hr);

    for (int k = 0; k < K; ++k)
    {
        free(fwd[k]);
        free(bwd[k]);
    }
    free(fwd);
    free(bwd);
    free(w);
    free(X);
    free(diff);
    free(a);
    free(b);
    free(tmp);
    free(reg);
    free(c);
    return 0;
}

/* BTV_L1_Base::process(src, dst, forwardMotions, baseIdx):
 * lib/src/btv_l1.cpp:328-355 */
ORC_API int orc_btvl1_process(const float* const* y, const float* const* fwd_motions, int K, int base_idx, int rows,
                              int cols, int ch, const orc_btv_params* p, float* out, float* x_full)
{
    const size_t cnt = (size_t)rows * cols * 2;
    float** rel = (float**)malloc(sizeof(float*) * K);
    for (int k = 0; k < K; ++k)
        rel[k] = (float*)malloc(sizeof(float) * cnt);
    orc_calc_relative_motions(fwd_motions, (float* const*)rel, K, base_idx, rows, cols);
    const int rc = orc_btvl1_run(y, (const float* const*)rel, K, base_idx, rows, cols, ch, p, out, x_full);
    for (int k = 0; k < K; ++k)
        free(rel[k]);
    free(rel);
    return rc;
}
