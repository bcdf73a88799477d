/*
 * CPU oracle for the built-in dense optical flow: duality-based TV-L1
 * (Zach, Pock, Bischof 2007; discretisation after Sanchez, Meinhardt-Llopis,
 * Facciolo, "TV-L1 Optical Flow Estimation", IPOL 2013, as also used by
 * OpenCV's DualTVL1).
 *
 * TEST INFRASTRUCTURE ONLY (see btvl1_oracle.c).
 *
 * PARITY UNPINNED: the reference contains no TV-L1 flow at all (its flow
 * classes wrap OpenCV's Farneback / SimpleFlow / Brox / PyrLK,
 * lib/src/optical_flow.cpp:86,151,229,290,358; TV-L1 is only cited,
 * lib/include/super_resolution.hpp:45).  This file is therefore the written
 * specification of the built-in flow; the GPU implementation is checked
 * against it bit for bit and against nothing else.
 *
 * Frozen discretisation
 *   input      frames as a source delivers them are reduced to 8-bit grey the way
 *              convertToType(frame, CV_8U, 1) does (lib/src/optical_flow.cpp:60-61,
 *              lib/src/input_array_utility.cpp:77-124): float -> round(v*255)
 *              saturated; 3/4 channels -> (B*1868 + G*9617 + R*4899 + 8192) >> 14;
 *              then converted to float.
 *   pyramid    factor 2: 5-tap binomial {1,4,6,4,1}, BORDER_REFLECT_101, even samples,
 *              size ((cols+1)/2, (rows+1)/2); a level smaller than 16 in either
 *              direction ends the pyramid (that level is not used).
 *   gradient   of I1: centred differences 0.5*(I(x+1)-I(x-1)), one-sided 0.5*(...) at
 *              the image edges.
 *   warping    bicubic (Keys, a = -0.75) at (x+u1, y+u2), horizontal taps then vertical,
 *              samples outside the image are 0.
 *   iteration  v = thresholding step; u = v + theta*div p with backward differences
 *              (first column/row: p itself); p = (p + tau/theta * grad u) /
 *              (1 + tau/theta * |grad u|) with forward differences (0 on the last
 *              column/row); all in float, in the order written below.
 *   stopping   an iteration's error is the sum over pixels of
 *              floor(min(du1^2 + du2^2, 4096) * 2^30) as a 64-bit integer (order
 *              independent, hence identical on any device); iterations continue while
 *              the error of the previous one exceeds floor(epsilon^2 * area * 2^30).
 *   upsampling flow to the next finer level: bilinear (cv::resize INTER_LINEAR rules),
 *              then times 2.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

int orc_border_reflect101(int p, int len);

typedef struct orc_tvl1_params
{
    double tau, lambda, theta;
    int nscales, warps;
    double epsilon;
    int iterations;
} orc_tvl1_params;

static int clampi(int v, int lo, int hi)
{
    return v < lo ? lo : v > hi ? hi : v;
}

/* frame (8U or 32F, 1/3/4 channels, dense) -> grey float in 0..255 */
ORC_API void orc_tvl1_to_gray(const void* src, int rows, int cols, int ch, int kind, float* dst)
{
    for (size_t i = 0; i < (size_t)rows * cols; ++i)
    {
        int c[4] = {0, 0, 0, 0};
        for (int k = 0; k < ch; ++k)
        {
            if (kind != 0) /* 1: float frames x 255; 2: 16-bit frames x 255/65535, scaled in float as cv::Mat::convertTo does */
            {
                const float v = kind == 1 ? ((const float*)src)[i * ch + k] * 255.0f
                                          : (float)((const unsigned short*)src)[i * ch + k] * (float)(255.0 / 65535.0);
                float r = nearbyintf(v);
                r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
                c[k] = (v != v) ? 0 : (int)r;
            }
            else
                c[k] = ((const unsigned char*)src)[i * ch + k];
        }
        dst[i] = ch == 1 ? (float)c[0] : (float)((c[0] * 1868 + c[1] * 9617 + c[2] * 4899 + 8192) >> 14);
    }
}

ORC_API void orc_pyr_down(const float* src, int rows, int cols, float* dst)
{
    const int dr = (rows + 1) / 2, dc = (cols + 1) / 2;
    float* tmp = (float*)malloc(sizeof(float) * (size_t)rows * dc);
    for (int y = 0; y < rows; ++y)
    {
        const float* S = src + (size_t)y * cols;
        for (int x = 0; x < dc; ++x)
        {
            const float s0 = S[orc_border_reflect101(2 * x - 2, cols)], s1 = S[orc_border_reflect101(2 * x - 1, cols)];
            const float s2 = S[2 * x], s3 = S[orc_border_reflect101(2 * x + 1, cols)];
            const float s4 = S[orc_border_reflect101(2 * x + 2, cols)];
            tmp[(size_t)y * dc + x] = s2 * 6 + (s1 + s3) * 4 + s0 + s4;
        }
    }
    for (int y = 0; y < dr; ++y)
    {
        const float* r0 = tmp + (size_t)orc_border_reflect101(2 * y - 2, rows) * dc;
        const float* r1 = tmp + (size_t)orc_border_reflect101(2 * y - 1, rows) * dc;
        const float* r2 = tmp + (size_t)(2 * y) * dc;
        const float* r3 = tmp + (size_t)orc_border_reflect101(2 * y + 1, rows) * dc;
        const float* r4 = tmp + (size_t)orc_border_reflect101(2 * y + 2, rows) * dc;
        for (int x = 0; x < dc; ++x)
            dst[(size_t)y * dc + x] = (r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x]) * 0.00390625f;
    }
    free(tmp);
}

ORC_API void orc_centered_gradient(const float* src, int rows, int cols, float* dx, float* dy)
{
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
        {
            const int xm = x > 0 ? x - 1 : 0, xp = x < cols - 1 ? x + 1 : cols - 1;
            const int ym = y > 0 ? y - 1 : 0, yp = y < rows - 1 ? y + 1 : rows - 1;
            dx[(size_t)y * cols + x] = 0.5f * (src[(size_t)y * cols + xp] - src[(size_t)y * cols + xm]);
            dy[(size_t)y * cols + x] = 0.5f * (src[(size_t)yp * cols + x] - src[(size_t)ym * cols + x]);
        }
}

static void cubic_coeffs(float x, float* c)
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

static float sample0(const float* img, int rows, int cols, int y, int x)
{
    return ((unsigned)x < (unsigned)cols && (unsigned)y < (unsigned)rows) ? img[(size_t)y * cols + x] : 0.0f;
}

/* bicubic sample of img at (fx, fy), zero outside */
static float warp_sample(const float* img, int rows, int cols, float fx, float fy)
{
    if (!(fabsf(fx) < 1e8f) || !(fabsf(fy) < 1e8f))
        return 0.0f;
    const float flx = floorf(fx), fly = floorf(fy);
    const int sx = (int)flx, sy = (int)fly;
    float wx[4], wy[4], h[4];
    cubic_coeffs(fx - flx, wx);
    cubic_coeffs(fy - fly, wy);
    for (int j = 0; j < 4; ++j)
        h[j] = sample0(img, rows, cols, sy - 1 + j, sx - 1) * wx[0] + sample0(img, rows, cols, sy - 1 + j, sx) * wx[1] +
               sample0(img, rows, cols, sy - 1 + j, sx + 1) * wx[2] +
               sample0(img, rows, cols, sy - 1 + j, sx + 2) * wx[3];
    return h[0] * wy[0] + h[1] * wy[1] + h[2] * wy[2] + h[3] * wy[3];
}

/* warp I1 and its gradient by (u1, u2); grad = |grad I1w|^2, rho_c = I1w - I1wx*u1 - I1wy*u2 - I0 */
ORC_API void orc_tvl1_warp(const float* I0, const float* I1, const float* I1x, const float* I1y, const float* u1,
                           const float* u2, int rows, int cols, float* I1w, float* I1wx, float* I1wy, float* grad,
                           float* rho_c)
{
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
        {
            const size_t i = (size_t)y * cols + x;
            const float fx = (float)x + u1[i], fy = (float)y + u2[i];
            const float w = warp_sample(I1, rows, cols, fx, fy);
            const float wx = warp_sample(I1x, rows, cols, fx, fy);
            const float wy = warp_sample(I1y, rows, cols, fx, fy);
            I1w[i] = w;
            I1wx[i] = wx;
            I1wy[i] = wy;
            grad[i] = wx * wx + wy * wy;
            rho_c[i] = w - wx * u1[i] - wy * u2[i] - I0[i];
        }
}

ORC_API uint64_t orc_tvl1_error_term(float du1, float du2)
{
    float q = du1 * du1 + du2 * du2;
    if (!(q < 4096.0f))
        q = 4096.0f; /* also catches NaN */
    return (uint64_t)(q * 1073741824.0f);
}

/* one inner iteration; returns the fixed-point error */
ORC_API uint64_t orc_tvl1_iteration(const float* I1wx, const float* I1wy, const float* grad, const float* rho_c,
                                    float* u1, float* u2, float* p11, float* p12, float* p21, float* p22, int rows,
                                    int cols, float l_t, float theta, float taut)
{
    uint64_t err = 0;
    /* thresholding step, divergence, primal update (in place: every value read here is u or p at
       already-final positions of the previous half step) */
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
        {
            const size_t i = (size_t)y * cols + x;
            const float rho = rho_c[i] + (I1wx[i] * u1[i] + I1wy[i] * u2[i]);
            float d1 = 0.0f, d2 = 0.0f;
            if (rho < -l_t * grad[i])
            {
                d1 = l_t * I1wx[i];
                d2 = l_t * I1wy[i];
            }
            else if (rho > l_t * grad[i])
            {
                d1 = -l_t * I1wx[i];
                d2 = -l_t * I1wy[i];
            }
            else if (grad[i] > FLT_EPSILON)
            {
                const float fi = -rho / grad[i];
                d1 = fi * I1wx[i];
                d2 = fi * I1wy[i];
            }
            const float v1 = u1[i] + d1, v2 = u2[i] + d2;
            const float div1 = (x > 0 ? p11[i] - p11[i - 1] : p11[i]) + (y > 0 ? p12[i] - p12[i - cols] : p12[i]);
            const float div2 = (x > 0 ? p21[i] - p21[i - 1] : p21[i]) + (y > 0 ? p22[i] - p22[i - cols] : p22[i]);
            const float n1 = v1 + theta * div1, n2 = v2 + theta * div2;
            err += orc_tvl1_error_term(n1 - u1[i], n2 - u2[i]);
            u1[i] = n1;
            u2[i] = n2;
        }
    /* forward gradient of u and dual update */
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
        {
            const size_t i = (size_t)y * cols + x;
            const float u1x = x < cols - 1 ? u1[i + 1] - u1[i] : 0.0f, u1y = y < rows - 1 ? u1[i + cols] - u1[i] : 0.0f;
            const float u2x = x < cols - 1 ? u2[i + 1] - u2[i] : 0.0f, u2y = y < rows - 1 ? u2[i + cols] - u2[i] : 0.0f;
            const float g1 = sqrtf(u1x * u1x + u1y * u1y), g2 = sqrtf(u2x * u2x + u2y * u2y);
            const float ng1 = 1.0f + taut * g1, ng2 = 1.0f + taut * g2;
            p11[i] = (p11[i] + taut * u1x) / ng1;
            p12[i] = (p12[i] + taut * u1y) / ng1;
            p21[i] = (p21[i] + taut * u2x) / ng2;
            p22[i] = (p22[i] + taut * u2y) / ng2;
        }
    return err;
}

/* cv::resize(INTER_LINEAR) on one float plane */
ORC_API void orc_resize_linear(const float* src, int srows, int scols, float* dst, int drows, int dcols)
{
    const double sx_ = 1.0 / ((double)dcols / scols), sy_ = 1.0 / ((double)drows / srows);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)srows * dcols);
    for (int dx = 0; dx < dcols; ++dx)
    {
        float fx = (float)((dx + 0.5) * sx_ - 0.5);
        int sx = (int)floor((double)fx);
        fx -= sx;
        if (sx < 0)
            fx = 0, sx = 0;
        if (sx >= scols - 1)
            fx = 0, sx = scols - 1;
        const float a0 = 1.f - fx, a1 = fx;
        for (int y = 0; y < srows; ++y)
        {
            const float* S = src + (size_t)y * scols;
            tmp[(size_t)y * dcols + dx] = sx >= scols - 1 ? S[scols - 1] : S[sx] * a0 + S[sx + 1] * a1;
        }
    }
    for (int dy = 0; dy < drows; ++dy)
    {
        float fy = (float)((dy + 0.5) * sy_ - 0.5);
        const int sy = (int)floor((double)fy);
        fy -= sy;
        const float b0 = 1.f - fy, b1 = fy;
        const float* S0 = tmp + (size_t)clampi(sy, 0, srows - 1) * dcols;
        const float* S1 = tmp + (size_t)clampi(sy + 1, 0, srows - 1) * dcols;
        for (int x = 0; x < dcols; ++x)
            dst[(size_t)dy * dcols + x] = S0[x] * b0 + S1[x] * b1;
    }
    free(tmp);
}

/* Flow I0 -> I1 of two grey float images; flow: rows x cols x 2 interleaved.
 * Returns the number of inner iterations executed. */
ORC_API long long orc_tvl1_calc(const float* I0, const float* I1, int rows, int cols, const orc_tvl1_params* p,
                                float* flow)
{
    enum
    {
        MAXS = 16
    };
    int nscales = p->nscales < 1 ? 1 : (p->nscales > MAXS ? MAXS : p->nscales);
    float *I0s[MAXS], *I1s[MAXS], *u1s[MAXS], *u2s[MAXS];
    int rs[MAXS], cs[MAXS];
    rs[0] = rows, cs[0] = cols;
    I0s[0] = (float*)malloc(sizeof(float) * (size_t)rows * cols);
    I1s[0] = (float*)malloc(sizeof(float) * (size_t)rows * cols);
    memcpy(I0s[0], I0, sizeof(float) * (size_t)rows * cols);
    memcpy(I1s[0], I1, sizeof(float) * (size_t)rows * cols);
    int built = 1;
    for (int s = 1; s < nscales; ++s)
    {
        rs[s] = (rs[s - 1] + 1) / 2, cs[s] = (cs[s - 1] + 1) / 2;
        if (cs[s] < 16 || rs[s] < 16)
        {
            nscales = s;
            break;
        }
        I0s[s] = (float*)malloc(sizeof(float) * (size_t)rs[s] * cs[s]);
        I1s[s] = (float*)malloc(sizeof(float) * (size_t)rs[s] * cs[s]);
        orc_pyr_down(I0s[s - 1], rs[s - 1], cs[s - 1], I0s[s]);
        orc_pyr_down(I1s[s - 1], rs[s - 1], cs[s - 1], I1s[s]);
        built = s + 1;
    }
    for (int s = 0; s < nscales; ++s)
    {
        u1s[s] = (float*)calloc((size_t)rs[s] * cs[s], sizeof(float));
        u2s[s] = (float*)calloc((size_t)rs[s] * cs[s], sizeof(float));
    }
    const float l_t = (float)(p->lambda * p->theta), taut = (float)(p->tau / p->theta), theta = (float)p->theta;
    long long executed = 0;
    for (int s = nscales - 1; s >= 0; --s)
    {
        const int R = rs[s], C = cs[s];
        const size_t n = (size_t)R * C;
        float* buf = (float*)calloc(11 * n, sizeof(float));
        float *I1x = buf, *I1y = buf + n, *I1w = buf + 2 * n, *I1wx = buf + 3 * n, *I1wy = buf + 4 * n;
        float *grad = buf + 5 * n, *rho_c = buf + 6 * n, *p11 = buf + 7 * n, *p12 = buf + 8 * n, *p21 = buf + 9 * n;
        float* p22 = buf + 10 * n;
        orc_centered_gradient(I1s[s], R, C, I1x, I1y);
        const float scaled_eps = (float)(p->epsilon * p->epsilon * (double)n);
        const uint64_t thr = (uint64_t)((double)scaled_eps * 1073741824.0);
        for (int w = 0; w < p->warps; ++w)
        {
            orc_tvl1_warp(I0s[s], I1s[s], I1x, I1y, u1s[s], u2s[s], R, C, I1w, I1wx, I1wy, grad, rho_c);
            uint64_t err = UINT64_MAX;
            for (int it = 0; err > thr && it < p->iterations; ++it)
            {
                err = orc_tvl1_iteration(I1wx, I1wy, grad, rho_c, u1s[s], u2s[s], p11, p12, p21, p22, R, C, l_t, theta,
                                         taut);
                ++executed;
            }
        }
        free(buf);
        if (s > 0)
        {
            orc_resize_linear(u1s[s], R, C, u1s[s - 1], rs[s - 1], cs[s - 1]);
            orc_resize_linear(u2s[s], R, C, u2s[s - 1], rs[s - 1], cs[s - 1]);
            const size_t m = (size_t)rs[s - 1] * cs[s - 1];
            for (size_t i = 0; i < m; ++i)
            {
                u1s[s - 1][i] = u1s[s - 1][i] * 2.0f;
                u2s[s - 1][i] = u2s[s - 1][i] * 2.0f;
            }
        }
    }
    for (size_t i = 0; i < (size_t)rows * cols; ++i)
    {
        flow[2 * i] = u1s[0][i];
        flow[2 * i + 1] = u2s[0][i];
    }
    for (int s = 0; s < nscales; ++s)
    {
        free(u1s[s]);
        free(u2s[s]);
    }
    for (int s = 0; s < built; ++s)
    {
        free(I0s[s]);
        free(I1s[s]);
    }
    return executed;
}
