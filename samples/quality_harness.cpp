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
//  * the low-resolution frames without noise: cv::GaussianBlur on 8-bit data (main.cpp:130-131) runs in fixed
//    point - kernel {1,4,6,4,1}/16 times 256 per pass, result (sum + 2^15) >> 16, i.e. the exact sum rounded half
//    up - and so does the blur here; INTER_NEAREST takes every scale-th pixel;
//  * the BiCubic column: cv::resize on 8-bit data (main.cpp:318) works with coefficients rounded to 11 bits and
//    integer sums, result (sum + 2^21) >> 22 saturated; restated here from OpenCV 2.4's resize (coefficients of
//    a = -0.75 in float, times 2048, rounded to nearest even; columns whose second tap falls outside the image
//    become plain copies, taps clipped to the image otherwise).
// Restated from the OpenCV 2.4 sources as the written spec, like the solver's OpenCV routines: nothing here was
// checked against a run of OpenCV (it is not installed), so "can be compared" means "is meant to agree".
// What cannot match to the digit:
//  * frames with --noise: the reference draws from cv::theRNG(), the harness from its own seeded generator.
#include <srx/superres.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

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

// AllignedFrameSource of samples/quality_test/main.cpp:43-75: a frame whose size is not a multiple of the scale is
// cut to the largest such size, anchored at the top-left corner (a view of the same memory, as Mat::operator()(Rect)).
class AlignedSource : public IFrameSource
{
  public:
    AlignedSource(Ptr<IFrameSource> base, int scale) : base_(base), scale_(scale) {}
    void reset() override { base_->reset(); }
    Mat nextFrame() override
    {
        Mat frame = base_->nextFrame();
        if (!frame.empty() && (frame.rows() % scale_ != 0 || frame.cols() % scale_ != 0))
            frame = frame.roi(0, 0, (frame.cols() / scale_) * scale_, (frame.rows() / scale_) * scale_);
        return frame;
    }

  private:
    Ptr<IFrameSource> base_;
    int scale_;
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
        // cv::GaussianBlur(8U, Size(5, 5), 0): both passes in integers with the kernel times 256
        static const int k[5] = {16, 64, 96, 64, 16};
        std::vector<int> tmp((size_t)R * C * ch);
        std::vector<uint8_t> blur((size_t)R * C * ch);
        for (int y = 0; y < R; ++y)
            for (int x = 0; x < C; ++x)
                for (int c = 0; c < ch; ++c)
                {
                    int s = 0;
                    for (int j = -2; j <= 2; ++j)
                        s += k[j + 2] * f.ptr<uint8_t>(y)[reflect101(x + j, C) * ch + c];
                    tmp[((size_t)y * C + x) * ch + c] = s;
                }
        for (int y = 0; y < R; ++y)
            for (int x = 0; x < C * ch; ++x)
            {
                int s = 0;
                for (int j = -2; j <= 2; ++j)
                    s += k[j + 2] * tmp[(size_t)reflect101(y + j, R) * C * ch + x];
                blur[(size_t)y * C * ch + x] = (uint8_t)((s + (1 << 15)) >> 16);
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
                    float v = (float)blur[((size_t)y * scale_ * C + (size_t)x * scale_) * ch + c];
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

// cv::resize(8U, INTER_CUBIC) by an integer factor: coefficients of a = -0.75 rounded to 11 bits, integer sums
short cubic_coeff_11(float v)
{
    const float r = std::nearbyint(v * 2048.0f); // cvRound: to nearest, ties to even
    return (short)std::fmin(std::fmax(r, -32768.f), 32767.f);
}
void cubic_table(int ssize, int dsize, bool snap, std::vector<int>& ofs, std::vector<short>& coef)
{
    ofs.resize(dsize), coef.resize(4 * (size_t)dsize);
    const double scale = 1.0 / ((double)dsize / ssize);
    for (int d = 0; d < dsize; ++d)
    {
        float f = (float)((d + 0.5) * scale - 0.5);
        int sp = (int)std::floor(f);
        f -= sp;
        if (snap) // columns: a second tap left of the image or on / behind its last column makes a plain copy
        {
            if (sp < 0)
                f = 0, sp = 0;
            if (sp >= ssize - 1)
                f = 0, sp = ssize - 1;
        }
        const float A = -0.75f;
        float c[4];
        c[0] = ((A * (f + 1) - 5 * A) * (f + 1) + 8 * A) * (f + 1) - 4 * A;
        c[1] = ((A + 2) * f - (A + 3)) * f * f + 1;
        c[2] = ((A + 2) * (1 - f) - (A + 3)) * (1 - f) * (1 - f) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
        ofs[d] = sp;
        for (int i = 0; i < 4; ++i)
            coef[4 * (size_t)d + i] = cubic_coeff_11(c[i]);
    }
}
Mat bicubic(const Mat& src, int scale)
{
    const int sr = src.rows(), sc = src.cols(), R = sr * scale, C = sc * scale, ch = src.channels();
    Mat dst(R, C, ch, SRX_8U);
    std::vector<int> xofs, yofs;
    std::vector<short> alpha, beta;
    cubic_table(sc, C, true, xofs, alpha);
    cubic_table(sr, R, false, yofs, beta);
    // every source row to the destination width
    std::vector<int> hres((size_t)sr * C * ch);
    for (int y = 0; y < sr; ++y)
        for (int x = 0; x < C; ++x)
            for (int c = 0; c < ch; ++c)
            {
                int s = 0;
                for (int i = 0; i < 4; ++i)
                    s += alpha[4 * (size_t)x + i] * src.ptr<uint8_t>(y)[std::min(std::max(xofs[x] - 1 + i, 0), sc - 1) * ch + c];
                hres[((size_t)y * C + x) * ch + c] = s;
            }
    for (int y = 0; y < R; ++y)
        for (int x = 0; x < C * ch; ++x)
        {
            int s = 0;
            for (int j = 0; j < 4; ++j)
                s += beta[4 * (size_t)y + j] * hres[(size_t)std::min(std::max(yofs[y] - 1 + j, 0), sr - 1) * C * ch + x];
            const int v = (s + (1 << 21)) >> 22;
            dst.ptr<uint8_t>(y)[x] = (uint8_t)std::min(std::max(v, 0), 255);
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

// --selftest: the host-side image routines against straightforward double-precision statements (no GPU involved)
int selftest()
{
    std::mt19937 rng(99);
    int bad = 0;
    for (int ch = 1; ch <= 3; ch += 2)
    {
        const int rows = 13, cols = 18, scale = 4;
        Mat img(rows, cols, ch, SRX_8U);
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols * ch; ++x)
                img.ptr<uint8_t>(y)[x] = (uint8_t)(rng() & 255);
        // bicubic: the fixed-point result is within one grey level of the same taps in double precision
        const Mat up = bicubic(img, scale);
        std::vector<int> xo, yo;
        std::vector<short> a, b;
        cubic_table(cols, cols * scale, true, xo, a);
        cubic_table(rows, rows * scale, false, yo, b);
        for (int y = 0; y < rows * scale; ++y)
            for (int x = 0; x < cols * scale; ++x)
                for (int c = 0; c < ch; ++c)
                {
                    double s = 0;
                    for (int j = 0; j < 4; ++j)
                        for (int i = 0; i < 4; ++i)
                            s += (b[4 * (size_t)y + j] / 2048.0) * (a[4 * (size_t)x + i] / 2048.0) *
                                 img.ptr<uint8_t>(std::min(std::max(yo[y] - 1 + j, 0), rows - 1))[std::min(std::max(xo[x] - 1 + i, 0), cols - 1) * ch + c];
                    const double want = std::fmin(std::fmax(std::floor(s + 0.5), 0.0), 255.0);
                    if (std::fabs(want - up.ptr<uint8_t>(y)[x * ch + c]) > 0.0)
                        ++bad;
                }
        // an image enlarged from a constant one stays within one level of it (the rounded coefficients need not add up to 2048)
        Mat flat(rows, cols, ch, SRX_8U);
        for (int y = 0; y < rows; ++y)
            std::memset(flat.ptr<uint8_t>(y), 200, (size_t)cols * ch);
        const Mat fu = bicubic(flat, scale);
        for (int y = 0; y < rows * scale; ++y)
            for (int x = 0; x < cols * scale * ch; ++x)
                if (std::abs((int)fu.ptr<uint8_t>(y)[x] - 200) > 1)
                    ++bad;
    }
    // degrade: exact sums of the 5x5 {1,4,6,4,1}/16 kernel rounded half up, then every scale-th pixel
    struct OneFrame : IFrameSource
    {
        Mat m;
        bool given = false;
        void reset() override { given = false; }
        Mat nextFrame() override
        {
            if (given)
                return Mat();
            given = true;
            return m;
        }
    };
    OneFrame* one = new OneFrame;
    one->m = Mat(12, 16, 1, SRX_8U);
    for (int y = 0; y < 12; ++y)
        for (int x = 0; x < 16; ++x)
            one->m.ptr<uint8_t>(y)[x] = (uint8_t)(rng() & 255);
    // two pixels arranged so that the exact sum of a window ends in .5: 255/256 * 128 ... checked below through the general rule
    DegradedSource deg(Ptr<IFrameSource>(one), 2, false);
    const Mat d = deg.nextFrame();
    static const double k[5] = {1 / 16.0, 4 / 16.0, 6 / 16.0, 4 / 16.0, 1 / 16.0};
    for (int y = 0; y < d.rows(); ++y)
        for (int x = 0; x < d.cols(); ++x)
        {
            double s = 0;
            for (int j = -2; j <= 2; ++j)
                for (int i = -2; i <= 2; ++i)
                    s += k[j + 2] * k[i + 2] * one->m.ptr<uint8_t>(reflect101(2 * y + j, 12))[reflect101(2 * x + i, 16)];
            if ((int)std::floor(s + 0.5) != d.ptr<uint8_t>(y)[x])
                ++bad;
        }
    // AlignedSource: a 13 x 18 clip at scale 4 becomes 12 x 16, the same pixels as the top-left corner of the source,
    // the degraded frame is 3 x 4, and a size that already fits passes through untouched
    {
        AlignedSource al(Ptr<IFrameSource>(new GoldSource(13, 18, 3, 2)), 4);
        GoldSource plain(13, 18, 3, 2);
        const Mat a = al.nextFrame(), g = plain.nextFrame();
        if (a.rows() != 12 || a.cols() != 16 || a.channels() != 3)
            ++bad;
        else
            for (int y = 0; y < 12; ++y)
                if (std::memcmp(a.ptr<uint8_t>(y), g.ptr<uint8_t>(y), 16 * 3) != 0)
                    ++bad;
        DegradedSource dal(Ptr<IFrameSource>(new AlignedSource(Ptr<IFrameSource>(new GoldSource(13, 18, 3, 2)), 4)), 4, false);
        DegradedSource dcut(Ptr<IFrameSource>(new GoldSource(12, 16, 3, 2)), 4, false);
        const Mat d1 = dal.nextFrame();
        if (d1.rows() != 3 || d1.cols() != 4)
            ++bad;
        // the cut comes first (main.cpp:269-271), so the blur reflects at the cut edge: the degraded frame equals that
        // of a clip generated at 12 x 16 (GoldSource's pixel values do not depend on the clip's size)
        const Mat d2 = dcut.nextFrame();
        for (int y = 0; y < 3 && d1.rows() == 3 && d1.cols() == 4; ++y)
            if (std::memcmp(d1.ptr<uint8_t>(y), d2.ptr<uint8_t>(y), 4 * 3) != 0)
                ++bad;
        AlignedSource fits(Ptr<IFrameSource>(new GoldSource(12, 16, 1, 1)), 4);
        const Mat f = fits.nextFrame();
        if (f.rows() != 12 || f.cols() != 16 || f.step() != 16)
            ++bad;
        if (!fits.nextFrame().empty())
            ++bad;
    }
    std::printf("selftest: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
    return bad ? 1 : 0;
}

int main(int argc, char** argv)
{
    if (arg_flag(argc, argv, "--selftest"))
        return selftest();
    const int scale = arg_int(argc, argv, "--scale", 4), iterations = arg_int(argc, argv, "--iterations", 180);
    const int temporal = arg_int(argc, argv, "--temporal", 4);
    // the clip may have any size: every source is wrapped in an AlignedSource, as at main.cpp:269-271
    const int rows = arg_int(argc, argv, "--rows", 240), cols = arg_int(argc, argv, "--cols", 320);
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

        auto aligned = [&]() {
            return Ptr<IFrameSource>(new AlignedSource(Ptr<IFrameSource>(new GoldSource(rows, cols, ch, frames)), scale));
        };
        Ptr<IFrameSource> gold = aligned();
        Ptr<IFrameSource> lowRes(new DegradedSource(aligned(), scale, noise));
        Ptr<IFrameSource> lowRes2(new DegradedSource(aligned(), scale, noise));
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

        std::printf("Input             : synthetic %dx%d (used: %dx%d), %d channel(s), %d frames%s\n", cols, rows,
                    cols / scale * scale, rows / scale * scale, ch, frames, noise ? ", noise" : "");
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
