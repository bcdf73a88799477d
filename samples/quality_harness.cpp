// Quality harness on the C++ facade: what samples/quality_test/main.cpp of the reference measures
// (super-resolved vs bicubic, PSNR and MSSIM against the ground truth), on a synthetic clip because
// no video decoder is linked here.
//
//   quality_harness [--scale 4] [--iterations 180] [--temporal 4] [--rows 240] [--cols 320]
//                   [--channels 1] [--frames 12] [--noise]
//
// Pipeline per frame (main.cpp:269-358): gold frame -> degraded frame (5x5 {1,4,6,4,1}/16 blur,
// REFLECT_101, then every scale-th pixel, main.cpp:130-131; optional Gaussian + spike noise,
// :133-134) -> BTV_L1::nextFrame(); metrics of the result against the gold frame cropped by
// btvKernelSize, and of a bicubic enlargement against the whole gold frame.
// As there, the first frame of the clip is dropped from all three sources before the loop (main.cpp:273-287).
//
// What can be compared with a run of the reference's program on the same clip:
//  * the metric functions: PSNR is exact (integer sum of squared differences), MSSIM follows the float
//    operations of main.cpp:140-199 one by one, including the colour case's read of component 3;
//  * the SuperRes column given the same low-resolution frames and the same motions: the solver is checked bit
//    for bit against the restatement of lib/src/btv_l1.cpp; with the built-in flows the motions differ from
//    OpenCV's in the last bits (Farneback and TV-L1 are restated from their published algorithms), so expect
//    agreement to a few hundredths of a dB, not to the digit.
// What cannot match to the digit:
//  * the low-resolution frames themselves: the reference degrades with cv::GaussianBlur on 8-bit data
//    (main.cpp:130-131), which runs in fixed point; the blur here runs in float and rounds once;
//  * the BiCubic column: cv::resize on 8-bit data (main.cpp:318) uses 11-bit fixed-point coefficients; here the
//    enlargement is computed in float and rounded.
#include <srx/superres.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

using namespace srx;

namespace
{

int reflect101(int p, int n)
{
    while (p < 0 || p >= n)
        p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// Ground truth: a band-limited texture under a slowly drifting camera.
class GoldSource : public IFrameSource
{
  public:
    GoldSource(int rows, int cols, int ch, int frames) : rows_(rows), cols_(cols), ch_(ch), frames_(frames) {}
    void reset() override { t_ = 0; }
    Mat nextFrame() override
    {
        if (t_ >= frames_)
            return Mat();
        Mat m(rows_, cols_, ch_, SRX_8U);
        const double tx = 1.3 * t_ + 0.6 * std::sin(0.9 * t_), ty = -0.7 * t_ + 0.8 * std::cos(0.7 * t_);
        for (int y = 0; y < rows_; ++y)
        {
            uint8_t* row = m.ptr<uint8_t>(y);
            for (int x = 0; x < cols_; ++x)
            {
                const double u = x + tx, v = y + ty;
                for (int c = 0; c < ch_; ++c)
                {
                    double s = 128 + 40 * std::sin(u / (5.0 + c)) * std::cos(v / 7.0) + 30 * std::sin((u + 2 * v) / 11.0) +
                               25 * std::sin((3 * u - v) / 17.0 + c);
                    s += (((int)std::floor(u / 24) + (int)std::floor(v / 24)) & 1) ? 18 : -18; // hard edges
                    row[x * ch_ + c] = (uint8_t)std::lround(std::fmin(std::fmax(s, 0.0), 255.0));
                }
            }
        }
        ++t_;
        return m;
    }

  private:
    int rows_, cols_, ch_, frames_, t_ = 0;
};

// blur + decimate (+ noise) of another source's frames
class DegradedSource : public IFrameSource
{
  public:
    DegradedSource(Ptr<IFrameSource> base, int scale, bool noise) : base_(base), scale_(scale), noise_(noise) {}
    void reset() override
    {
        base_->reset();
        rng_.seed(1234);
    }
    Mat nextFrame() override
    {
        Mat f = base_->nextFrame();
        if (f.empty())
            return f;
        const int R = f.rows(), C = f.cols(), ch = f.channels();
        static const float k[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
        std::vector<float> tmp((size_t)R * C * ch), blur((size_t)R * C * ch);
        for (int y = 0; y < R; ++y)
            for (int x = 0; x < C; ++x)
                for (int c = 0; c < ch; ++c)
                {
                    float s = 0;
                    for (int j = -2; j <= 2; ++j)
                        s += k[j + 2] * f.ptr<uint8_t>(y)[reflect101(x + j, C) * ch + c];
                    tmp[((size_t)y * C + x) * ch + c] = s;
                }
        for (int y = 0; y < R; ++y)
            for (int x = 0; x < C * ch; ++x)
            {
                float s = 0;
                for (int j = -2; j <= 2; ++j)
                    s += k[j + 2] * tmp[(size_t)reflect101(y + j, R) * C * ch + x];
                blur[(size_t)y * C * ch + x] = s;
            }
        Mat d(R / scale_, C / scale_, ch, SRX_8U);
        std::normal_distribution<float> gauss(0.f, 10.f);
        std::uniform_int_distribution<int> spike(0, 499);
        for (int y = 0; y < d.rows(); ++y)
            for (int x = 0; x < d.cols(); ++x)
            {
                const bool hit = noise_ && spike(rng_) < 1;
                for (int c = 0; c < ch; ++c)
                {
                    float v = std::nearbyint(blur[((size_t)y * scale_ * C + (size_t)x * scale_) * ch + c]);
                    if (noise_)
                        v = std::nearbyint(v + gauss(rng_));
                    v = hit ? 255.f : std::fmin(std::fmax(v, 0.f), 255.f);
                    d.ptr<uint8_t>(y)[x * ch + c] = (uint8_t)v;
                }
            }
        return d;
    }

  private:
    Ptr<IFrameSource> base_;
    int scale_;
    bool noise_;
    std::mt19937 rng_{1234};
};

// cubic enlargement in float (a = -0.75, taps clamped to the image), rounded to 8 bits
Mat bicubic(const Mat& src, int scale)
{
    const int R = src.rows() * scale, C = src.cols() * scale, ch = src.channels();
    Mat dst(R, C, ch, SRX_8U);
    auto w = [](float t, float* o) {
        const float A = -0.75f;
        o[0] = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A;
        o[1] = ((A + 2) * t - (A + 3)) * t * t + 1;
        o[2] = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1;
        o[3] = 1.f - o[0] - o[1] - o[2];
    };
    for (int y = 0; y < R; ++y)
    {
        const float fy = (y + 0.5f) / scale - 0.5f;
        const int sy = (int)std::floor(fy);
        float wy[4];
        w(fy - sy, wy);
        for (int x = 0; x < C; ++x)
        {
            const float fx = (x + 0.5f) / scale - 0.5f;
            const int sx = (int)std::floor(fx);
            float wx[4];
            w(fx - sx, wx);
            for (int c = 0; c < ch; ++c)
            {
                float s = 0;
                for (int j = 0; j < 4; ++j)
                {
                    const int yy = std::min(std::max(sy - 1 + j, 0), src.rows() - 1);
                    float r = 0;
                    for (int i = 0; i < 4; ++i)
                        r += wx[i] * src.ptr<uint8_t>(yy)[std::min(std::max(sx - 1 + i, 0), src.cols() - 1) * ch + c];
                    s += wy[j] * r;
                }
                dst.ptr<uint8_t>(y)[x * ch + c] = (uint8_t)std::fmin(std::fmax(std::nearbyint(s), 0.f), 255.f);
            }
        }
    }
    return dst;
}

int arg_int(int argc, char** argv, const char* name, int dflt)
{
    for (int i = 1; i + 1 < argc; ++i)
        if (std::strcmp(argv[i], name) == 0)
            return std::atoi(argv[i + 1]);
    return dflt;
}
const char* arg_str(int argc, char** argv, const char* name, const char* dflt)
{
    for (int i = 1; i + 1 < argc; ++i)
        if (std::strcmp(argv[i], name) == 0)
            return argv[i + 1];
    return dflt;
}

bool arg_flag(int argc, char** argv, const char* name)
{
    for (int i = 1; i < argc; ++i)
        if (std::strcmp(argv[i], name) == 0)
            return true;
    return false;
}

} // namespace

int main(int argc, char** argv)
{
    const int scale = arg_int(argc, argv, "--scale", 4), iterations = arg_int(argc, argv, "--iterations", 180);
    const int temporal = arg_int(argc, argv, "--temporal", 4);
    const int rows = arg_int(argc, argv, "--rows", 240) / scale * scale, cols = arg_int(argc, argv, "--cols", 320) / scale * scale;
    const int ch = arg_int(argc, argv, "--channels", 1), frames = arg_int(argc, argv, "--frames", 12);
    const bool noise = arg_flag(argc, argv, "--noise");
    // --opt-flow farneback | tvl1; farneback is the default of samples/quality_test/main.cpp:209
    const std::string optFlow = arg_str(argc, argv, "--opt-flow", "farneback");
    if (optFlow != "farneback" && optFlow != "tvl1")
    {
        std::fprintf(stderr, "Incorrect Optical Flow algorithm - %s\n", optFlow.c_str());
        return 1;
    }
    try
    {
        // the reference's program does `new BTV_L1` / `new BTV_L1_GPU` (main.cpp:255-259); creation by the
        // registered name reaches the same class
        if (!initModule_superres())
            return 1;
        Ptr<SuperResolution> superRes = Algorithm::create<SuperResolution>("SuperResolution.BTV_L1");
        const int B = superRes->getInt("btvKernelSize");
        superRes->set("scale", scale);
        superRes->set("iterations", iterations);
        superRes->set("temporalAreaRadius", temporal);
        superRes->setOpticalFlow(optFlow == "tvl1" ? createOptFlow_DualTVL1() : createOptFlow_Farneback());

        Ptr<IFrameSource> gold(new GoldSource(rows, cols, ch, frames));
        Ptr<IFrameSource> lowRes(new DegradedSource(Ptr<IFrameSource>(new GoldSource(rows, cols, ch, frames)), scale, noise));
        Ptr<IFrameSource> lowRes2(new DegradedSource(Ptr<IFrameSource>(new GoldSource(rows, cols, ch, frames)), scale, noise));
        // "skip first frame, it is usually corrupted" (main.cpp:273-287): all three sources drop their first
        // frame before the super-resolution object sees its source, so the table starts at the clip's second frame
        {
            Mat frame = lowRes->nextFrame();
            frame = lowRes2->nextFrame();
            frame = gold->nextFrame();
            if (frame.empty())
                return 2;
        }
        superRes->setFrameSource(lowRes);

        std::printf("Input             : synthetic %dx%d, %d channel(s), %d frames%s\n", cols, rows, ch, frames,
                    noise ? ", noise" : "");
        std::printf("Scale factor      : %d\nIterations        : %d\nFrames to process : %d\nOptical Flow      : %s\n"
                    "Mode              : GPU\n\n",
                    scale, iterations, 2 * temporal + 1, optFlow.c_str());
        std::printf("-----------------------------------------------------------------\n");
        std::printf("|   Ind    |  SuperRes PSNR / MSSIM  |   BiCubic PSNR / MSSIM   |\n");
        std::printf("|----------|-------------------------|--------------------------|\n");
        double srP = 0, srM = 0, bcP = 0, bcM = 0;
        int count = 0;
        for (;;)
        {
            Mat goldFrame = gold->nextFrame();
            if (goldFrame.empty())
                break;
            Mat lowResFrame = lowRes2->nextFrame();
            if (lowResFrame.empty())
                break;
            Mat superResFrame = superRes->nextFrame();
            if (superResFrame.empty())
                break;
            Mat biCubicFrame = bicubic(lowResFrame, scale);
            Mat inner = goldFrame.roi(B, B, goldFrame.cols() - 2 * B, goldFrame.rows() - 2 * B);
            const double p1 = PSNR(inner, superResFrame), m1 = MSSIM(inner, superResFrame);
            const double p2 = PSNR(goldFrame, biCubicFrame), m2 = MSSIM(goldFrame, biCubicFrame);
            srP += p1, srM += m1, bcP += p2, bcM += m2;
            ++count;
            std::printf("|  [%4d]  |      %.2f / %.3f      |       %.2f / %.3f      |\n", count, p1, m1, p2, m2);
        }
        std::printf("-----------------------------------------------------------------\n\n");
        if (count == 0)
            return 2;
        std::printf("Super Resolution Avg PSNR  : %.4f\nSuper Resolution Avg MSSIM : %.5f\n\n", srP / count, srM / count);
        std::printf("Bi-Cubic Resize  AVG PSNR  : %.4f\nBi-Cubic Resize  AVG MSSIM : %.5f\n", bcP / count, bcM / count);
        return srP > bcP ? 0 : 3; // a super-resolved clip that does not beat bicubic is a failure
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "quality_harness: %s\n", e.what());
        return 1;
    }
}
