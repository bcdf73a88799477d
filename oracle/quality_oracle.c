/*
 * CPU oracle for the quality metrics of the reference's quality harness
 * (samples/quality_test/main.cpp): PSNR, MSSIM, and the 8-bit bicubic resize of
 * the comparison column.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ as the checker of the GPU metrics;
 * nothing in the shipped GPU path links, loads or calls it.
 *
 * PARITY UNPINNED: the reference ships no expected metric values and OpenCV is
 * not available, so cv::PSNR, cv::GaussianBlur, the MatExpr arithmetic, cv::mean
 * and cv::resize(8U, INTER_CUBIC) are restated from the OpenCV 2.4 sources as the
 * written spec.  Build with -ffp-contract=off.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* btvl1_oracle.c */
void orc_gaussian_kernel(int n, double sigma, float* k);
void orc_sep_filter(const float* src, int rows, int cols, int ch, const float* k, int n, float* tmp, float* dst);

/* cv::PSNR (samples/quality_test/main.cpp:320,323): 8-bit inputs only,
 *   diff = sqrt(norm(a, b, NORM_L2SQR) / (total * channels));  20 * log10(255 / (diff + DBL_EPSILON))
 * The squared differences of bytes sum exactly in integers.  Images are given
 * with a row pitch in bytes so that a region of interest can be passed. */
ORC_API double orc_psnr_8u(const uint8_t* a, size_t a_step, const uint8_t* b, size_t b_step, int rows, int cols, int ch)
{
    uint64_t s = 0;
    for (int y = 0; y < rows; ++y)
    {
        const uint8_t* pa = a + (size_t)y * a_step;
        const uint8_t* pb = b + (size_t)y * b_step;
        for (int x = 0; x < cols * ch; ++x)
        {
            const int d = (int)pa[x] - (int)pb[x];
            s += (uint64_t)(d * d);
        }
    }
    const double diff = sqrt((double)s / ((double)rows * cols * ch));
    return 20.0 * log10(255.0 / (diff + DBL_EPSILON));
}

/* MSSIM (samples/quality_test/main.cpp:140-199) on float images (the caller
 * has done convertTo(CV_32F), :147-149), channels interleaved, dense.
 *   mean_out[c]  mean of the SSIM map of channel c (cv::mean, double sums in
 *                row-major order), c < ch; the rest 0
 * Returns what the reference returns: mssim[0] for one channel, otherwise
 * (mssim[0] + mssim[1] + mssim[3]) / 3 -- the reference reads component 3 where
 * 2 is meant (:198); for a 3-channel image component 3 is 0. */
ORC_API double orc_mssim(const float* I1, const float* I2, int rows, int cols, int ch, double* mean_out)
{
    const float C1 = (float)6.5025, C2 = (float)58.5225; /* Scalar -> float for CV_32F arithmetic */
    const size_t n = (size_t)rows * cols * ch;
    float k[11];
    orc_gaussian_kernel(11, 1.5, k); /* GaussianBlur(..., Size(11, 11), 1.5), border REFLECT_101 */

    float* I1_2 = (float*)malloc(sizeof(float) * n);
    float* I2_2 = (float*)malloc(sizeof(float) * n);
    float* I1_I2 = (float*)malloc(sizeof(float) * n);
    float* mu1 = (float*)malloc(sizeof(float) * n);
    float* mu2 = (float*)malloc(sizeof(float) * n);
    float* s11 = (float*)malloc(sizeof(float) * n);
    float* s22 = (float*)malloc(sizeof(float) * n);
    float* s12 = (float*)malloc(sizeof(float) * n);
    float* tmp = (float*)malloc(sizeof(float) * n);

    for (size_t e = 0; e < n; ++e) /* :151-153 */
    {
        I2_2[e] = I2[e] * I2[e];
        I1_2[e] = I1[e] * I1[e];
        I1_I2[e] = I1[e] * I2[e];
    }
    orc_sep_filter(I1, rows, cols, ch, k, 11, tmp, mu1);    /* :156 */
    orc_sep_filter(I2, rows, cols, ch, k, 11, tmp, mu2);    /* :157 */
    orc_sep_filter(I1_2, rows, cols, ch, k, 11, tmp, s11);  /* :165 */
    orc_sep_filter(I2_2, rows, cols, ch, k, 11, tmp, s22);  /* :168 */
    orc_sep_filter(I1_I2, rows, cols, ch, k, 11, tmp, s12); /* :171 */

    /* cv::mean = cv::sum / N.  sum_<float, double>: one channel is summed four elements at a time,
     * the four added in float before they join the double total (CV_ENABLE_UNROLLED); with three
     * channels every element joins its channel's double total on its own. */
    double sum[4] = {0, 0, 0, 0};
    float* map = tmp; /* the SSIM map */
    for (size_t e = 0; e < n; ++e)
    {
        const float mu1_2 = mu1[e] * mu1[e];     /* :159 */
        const float mu2_2 = mu2[e] * mu2[e];     /* :160 */
        const float mu1_mu2 = mu1[e] * mu2[e];   /* :161 */
        const float sigma1_2 = s11[e] - mu1_2;   /* :166 */
        const float sigma2_2 = s22[e] - mu2_2;   /* :169 */
        const float sigma12 = s12[e] - mu1_mu2;  /* :172 */
        /* :179-181  convertTo(scale 2, shift C) then mul */
        float t1 = mu1_mu2 * 2.0f + C1;
        float t2 = sigma12 * 2.0f + C2;
        const float numerator = t1 * t2;
        /* :184-186  add, add scalar, mul */
        t1 = (mu1_2 + mu2_2) + C1;
        t2 = (sigma1_2 + sigma2_2) + C2;
        const float denominator = t1 * t2;
        /* :190 cv::divide: 0 where the divisor is 0 */
        const float ssim = denominator != 0.0f ? (float)((double)numerator * 1.0 / (double)denominator) : 0.0f;
        map[e] = ssim;
    }
    if (ch == 1)
    {
        size_t e = 0;
        for (; e + 4 <= n; e += 4)
            sum[0] += map[e] + map[e + 1] + map[e + 2] + map[e + 3];
        for (; e < n; ++e)
            sum[0] += map[e];
    }
    else
        for (size_t e = 0; e < n; ++e)
            sum[e % ch] += (double)map[e];
    double m[4] = {0, 0, 0, 0};
    for (int c = 0; c < ch && c < 4; ++c)
        m[c] = sum[c] / ((double)rows * cols); /* :193 */
    if (mean_out)
        memcpy(mean_out, m, sizeof(m));

    free(I1_2), free(I2_2), free(I1_I2), free(mu1), free(mu2), free(s11), free(s22), free(s12), free(tmp);
    if (ch == 1)
        return m[0];
    return (m[0] + m[1] + m[3]) / 3; /* :198, as written in the reference */
}
