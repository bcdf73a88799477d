// Streams a recorded clip through the C++ facade (include/srx/superres.hpp) on a GPU and compares every
// delivered frame, byte for byte, with the recorded outputs.  Driven by tests/test_gpu_cpp_facade.py, which
// writes the clip of tests/golden/stream_rgb_r2.npz as raw files into a directory:
//   meta.txt   n rows cols channels radius iterations out_rows out_cols
//   frames.bin n frames, 8-bit          flows.bin  n-1 forward motions (rows x cols x 2 float)
//   outs.bin   n expected output frames, 8-bit
// The object is made the way the reference's programs make it (samples/quality_test/main.cpp:255-267):
// initModule_superres(), Algorithm::create<SuperResolution>("SuperResolution.BTV_L1"), set(name, value).
#include <srx/superres.hpp>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#define REQUIRE(c)                                                                                                     \
    do                                                                                                                 \
    {                                                                                                                  \
        if (!(c))                                                                                                      \
        {                                                                                                              \
            std::fprintf(stderr, "%s:%d: requirement failed: %s\n", __FILE__, __LINE__, #c);                           \
            std::exit(1);                                                                                              \
        }                                                                                                              \
    } while (0)

using namespace srx;

static std::vector<uint8_t> slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    REQUIRE(f.good());
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

struct Clip
{
    int n = 0, rows = 0, cols = 0, ch = 0, radius = 0, iterations = 0, orows = 0, ocols = 0;
    std::vector<uint8_t> frames, flows, outs;
    Mat frame(int i) { return Mat(rows, cols, ch, SRX_8U, frames.data() + (size_t)i * rows * cols * ch); }
    Mat flow(int i) { return Mat(rows, cols, 2, SRX_32F, flows.data() + (size_t)i * rows * cols * 2 * sizeof(float)); }
    const uint8_t* out(int i) const { return outs.data() + (size_t)i * orows * ocols * ch; }
};

struct ClipSource : IFrameSource
{
    Clip* clip;
    int pos = 0;
    explicit ClipSource(Clip* c) : clip(c) {}
    void reset() override { pos = 0; }
    Mat nextFrame() override { return pos < clip->n ? clip->frame(pos++) : Mat(); }
};

// hands out the recorded motions in order: one interleaved image, or (planar) two planes u and v
struct RecordedFlow : DenseOpticalFlow
{
    Clip* clip;
    bool two_outputs;
    int pos = 0;
    RecordedFlow(Clip* c, bool two) : clip(c), two_outputs(two) {}
    bool planar() const override { return two_outputs; }
    void calc(const Mat& f0, const Mat&, Mat& flow1, Mat* flow2) override
    {
        REQUIRE(pos < clip->n - 1 && (flow2 != nullptr) == two_outputs);
        const float* src = (const float*)clip->flow(pos++).ptr<float>();
        const size_t px = (size_t)f0.rows() * f0.cols();
        if (!flow2)
        {
            flow1.create(f0.rows(), f0.cols(), 2, SRX_32F);
            std::memcpy(flow1.ptr<float>(), src, px * 2 * sizeof(float));
            return;
        }
        flow1.create(f0.rows(), f0.cols(), 1, SRX_32F);
        flow2->create(f0.rows(), f0.cols(), 1, SRX_32F);
        for (size_t i = 0; i < px; ++i)
            flow1.ptr<float>()[i] = src[2 * i], flow2->ptr<float>()[i] = src[2 * i + 1];
    }
};

static void stream_and_compare(Clip& clip, bool two_outputs, bool by_member)
{
    Ptr<SuperResolution> sr = Algorithm::create<SuperResolution>("SuperResolution.BTV_L1");
    REQUIRE(sr);
    sr->set("iterations", clip.iterations);
    if (by_member)
        std::dynamic_pointer_cast<BTV_L1>(sr)->temporalAreaRadius = clip.radius;
    else
        sr->set("temporalAreaRadius", clip.radius);
    auto flow = std::make_shared<RecordedFlow>(&clip, two_outputs);
    sr->set("opticalFlow", Ptr<DenseOpticalFlow>(flow));
    sr->setFrameSource(std::make_shared<ClipSource>(&clip));
    const size_t obytes = (size_t)clip.orows * clip.ocols * clip.ch;
    for (int i = 0; i < clip.n; ++i)
    {
        Mat f = sr->nextFrame();
        REQUIRE(!f.empty() && f.rows() == clip.orows && f.cols() == clip.ocols && f.channels() == clip.ch);
        REQUIRE(f.depth() == SRX_8U && !f.onDevice() && f.step() == (size_t)clip.ocols * clip.ch);
        if (std::memcmp(f.ptr<uint8_t>(), clip.out(i), obytes) != 0)
        {
            std::fprintf(stderr, "output frame %d differs from the recorded one (two_outputs=%d)\n", i, (int)two_outputs);
            std::exit(1);
        }
    }
    REQUIRE(sr->nextFrame().empty());
    REQUIRE(flow->pos == clip.n - 1);
}

int main(int argc, char** argv)
{
    REQUIRE(argc == 2);
    const std::string dir = argv[1];
    Clip clip;
    {
        std::ifstream m(dir + "/meta.txt");
        REQUIRE(m.good());
        m >> clip.n >> clip.rows >> clip.cols >> clip.ch >> clip.radius >> clip.iterations >> clip.orows >> clip.ocols;
        REQUIRE(m.good() || m.eof());
    }
    clip.frames = slurp(dir + "/frames.bin");
    clip.flows = slurp(dir + "/flows.bin");
    clip.outs = slurp(dir + "/outs.bin");
    REQUIRE(clip.frames.size() == (size_t)clip.n * clip.rows * clip.cols * clip.ch);
    REQUIRE(clip.flows.size() == (size_t)(clip.n - 1) * clip.rows * clip.cols * 2 * sizeof(float));
    REQUIRE(clip.outs.size() == (size_t)clip.n * clip.orows * clip.ocols * clip.ch);

    REQUIRE(initModule_superres());
    stream_and_compare(clip, false, false); // calc(f0, f1, flow), radius by name
    stream_and_compare(clip, true, true);   // calc(f0, f1, u, v), radius through the public member

    // BTV_L1_Base::process with forward motions on the window of output frame `radius` (full window, base in
    // the middle: lib/src/btv_l1.cpp:527-549 hands the same frames and motions to run()) gives that frame
    {
        BTV_L1_Base base;
        base.iterations = clip.iterations;
        const int r = clip.radius;
        REQUIRE(clip.n >= 2 * r + 1);
        std::vector<Mat> src, motions;
        for (int i = 0; i <= 2 * r; ++i)
            src.push_back(clip.frame(i));
        for (int i = 0; i < 2 * r; ++i)
            motions.push_back(clip.flow(i));
        Mat dst;
        base.process(src, dst, motions, r);
        REQUIRE(dst.rows() == clip.orows && dst.cols() == clip.ocols && dst.depth() == SRX_8U);
        REQUIRE(std::memcmp(dst.ptr<uint8_t>(), clip.out(r), (size_t)clip.orows * clip.ocols * clip.ch) == 0);
    }
    std::puts("facade stream ok");
    return 0;
}
