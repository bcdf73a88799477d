// Exercises the C++ facade (include/srx/superres.hpp) on a dry-run object: no GPU needed.
// Built and run by tests/test_cpp_facade.py.
#include <srx/superres.hpp>

#include <cstdio>
#include <cstdlib>

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

struct CountingSource : IFrameSource
{
    int n, pos = 0, resets = 0;
    Mat frame;
    explicit CountingSource(int n) : n(n), frame(6, 8, 3, SRX_8U) {}
    void reset() override { pos = 0, ++resets; }
    Mat nextFrame() override { return pos++ < n ? frame : Mat(); }
};

struct ExtSource : FrameSource
{
    int n, pos = 0;
    Mat frame;
    explicit ExtSource(int n) : n(n), frame(6, 8, 1, SRX_8U) {}
    void nextFrame(Mat& f) override { f = pos++ < n ? frame : Mat(); }
    void reset() override { pos = 0; }
};

struct ZeroFlow : DenseOpticalFlow
{
    int calls = 0;
    void calc(const Mat& f0, const Mat& f1, Mat& flow1, Mat*) override
    {
        REQUIRE(f0.rows() == f1.rows() && f0.cols() == f1.cols());
        flow1.create(f0.rows(), f0.cols(), 2, SRX_32F);
        ++calls;
    }
};

// the two-output form: planar u and v
struct PlanarZeroFlow : DenseOpticalFlow
{
    int calls = 0;
    bool planar() const override { return true; }
    void calc(const Mat& f0, const Mat&, Mat& u, Mat* v) override
    {
        REQUIRE(v != nullptr);
        u.create(f0.rows(), f0.cols(), 1, SRX_32F);
        v->create(f0.rows(), f0.cols(), 1, SRX_32F);
        ++calls;
    }
};

int main()
{
    // creation by registered name, as after the reference's initModule_superres()
    // (lib/src/super_resolution.cpp:83-123)
    Algorithm::defaultDevice() = -1;
    REQUIRE(!Algorithm::create<SuperResolution>("SuperResolution.BTV_L1")); // not registered yet
    REQUIRE(initModule_superres());
    REQUIRE(initModule_superres()); // a second call registers nothing twice
    {
        std::vector<std::string> names;
        Algorithm::getList(names);
        REQUIRE(names.size() == 2 && names[0] == "SuperResolution.BTV_L1" && names[1] == "SuperResolution.BTV_L1_GPU");
        Ptr<SuperResolution> made = Algorithm::create<SuperResolution>("SuperResolution.BTV_L1");
        REQUIRE(made && made->name() == "SuperResolution.BTV_L1");
        REQUIRE(made->getInt("scale") == 4 && made->getInt("temporalAreaRadius") == 4);
        made->set("scale", 2);
        made->set("temporalAreaRadius", 1);
        auto planar = std::make_shared<PlanarZeroFlow>();
        made->set("opticalFlow", Ptr<DenseOpticalFlow>(planar));
        made->setFrameSource(std::make_shared<CountingSource>(5));
        int outs = 0;
        while (!made->nextFrame().empty())
            ++outs;
        REQUIRE(outs == 5 && planar->calls == 4);
        REQUIRE(!Algorithm::create<SuperResolution>("SuperResolution.Unknown"));
        REQUIRE(!Algorithm::create<DenseOpticalFlow>("SuperResolution.BTV_L1")); // another type: empty, as cv::Algorithm
    }

    // BTV_L1_Base on its own (lib/include/super_resolution.hpp:46-62): public members, both process overloads
    {
        BTV_L1_Base base(-1);
        REQUIRE(base.scale == 4 && base.iterations == 180 && base.tau == 1.3 && base.lambda == 0.03);
        REQUIRE(base.alpha == 0.7 && base.btvKernelSize == 7 && base.blurKernelSize == 5 && base.blurSigma == 0.0);
        REQUIRE(dynamic_cast<Farneback*>(base.opticalFlow.get()) != nullptr);
        base.scale = 2;
        base.btvKernelSize = 5;
        auto zf = std::make_shared<ZeroFlow>();
        base.opticalFlow = zf;
        std::vector<Mat> src(3, Mat(12, 16, 1, SRX_8U));
        Mat dst;
        base.process(src, dst, 1); // motions from opticalFlow: one call per frame other than the base
        REQUIRE(zf->calls == 2);
        REQUIRE(dst.rows() == 12 * 2 - 10 && dst.cols() == 16 * 2 - 10 && dst.channels() == 1);
        std::vector<Mat> motions(2, Mat(12, 16, 2, SRX_32F));
        Mat dst2;
        base.process(src, dst2, motions, 1);
        REQUIRE(zf->calls == 2 && dst2.rows() == dst.rows());
        int scale_in_library = 0;
        REQUIRE(srx_get_int(base.handle(), "scale", &scale_in_library) == SRX_OK && scale_in_library == 2);
        base.iterations = 0; // the reference debug-asserts iterations > 0 (lib/src/btv_l1.cpp:361)
        bool threw = false;
        try
        {
            base.process(src, dst2, motions, 1);
        }
        catch (const Exception& e)
        {
            threw = e.code == SRX_ERR_INVALID;
        }
        REQUIRE(threw);
    }
    // reference spelling: new BTV_L1, set/getInt, setFrameSource, nextFrame (samples/quality_test/main.cpp:255-316)
    BTV_L1 sr(-1);
    REQUIRE(sr.getInt("scale") == 4 && sr.getInt("iterations") == 180 && sr.getInt("btvKernelSize") == 7);
    REQUIRE(sr.getInt("blurKernelSize") == 5 && sr.getInt("temporalAreaRadius") == 4);
    REQUIRE(sr.getDouble("tau") == 1.3 && sr.getDouble("lambda") == 0.03 && sr.getDouble("alpha") == 0.7);
    REQUIRE(sr.getDouble("blurSigma") == 0.0);
    sr.set("temporalAreaRadius", 2);
    sr.set("iterations", 20);
    auto flow = std::make_shared<ZeroFlow>();
    sr.set("opticalFlow", Ptr<DenseOpticalFlow>(flow));
    auto src = std::make_shared<CountingSource>(9);
    sr.setFrameSource(src);
    int outs = 0;
    for (;;)
    {
        Mat f = sr.nextFrame();
        if (f.empty())
            break;
        REQUIRE(f.rows() == 6 * 4 - 14 && f.cols() == 8 * 4 - 14 && f.channels() == 3 && f.depth() == SRX_8U);
        ++outs;
    }
    REQUIRE(outs == 9);
    REQUIRE(flow->calls == 8); // one flow per consecutive frame pair
    sr.reset();
    REQUIRE(src->resets == 1);
    REQUIRE(!sr.nextFrame().empty());

    // invalid parameters throw instead of the reference's release-mode undefined behaviour
    bool threw = false;
    try
    {
        sr.set("scale", 1);
    }
    catch (const Exception& e)
    {
        threw = e.code == SRX_ERR_INVALID;
    }
    REQUIRE(threw && sr.getInt("scale") == 4);

    // temporalAreaRadius is a public member (lib/include/super_resolution.hpp:172), read when the stream starts
    {
        BTV_L1 direct(-1);
        direct.temporalAreaRadius = 1;
        REQUIRE(direct.getInt("temporalAreaRadius") == 1);
        direct.setOpticalFlow(flow);
        auto few = std::make_shared<CountingSource>(3); // 2r+1 frames are enough for radius 1, not for radius 4
        direct.setFrameSource(few);
        int outs = 0;
        while (!direct.nextFrame().empty())
            ++outs;
        REQUIRE(outs == 3);
        direct.set("temporalAreaRadius", 2);
        REQUIRE(direct.temporalAreaRadius == 2);
    }

    // a stream shorter than 2r+1 frames trips the warm-up assertion (lib/src/btv_l1.cpp:479)
    threw = false;
    BTV_L1 shortsr(-1);
    shortsr.setOpticalFlow(flow);
    shortsr.setFrameSource(std::make_shared<CountingSource>(5));
    try
    {
        shortsr.nextFrame();
    }
    catch (const Exception& e)
    {
        threw = e.code == SRX_ERR_SOURCE;
    }
    REQUIRE(threw);

    // upstream spelling: createSuperResolution_BTVL1, setInput, nextFrame(Mat&), typed accessors
    Ptr<BTV_L1> up = createSuperResolution_BTVL1(-1);
    up->setScale(2);
    up->setTemporalAreaRadius(1);
    up->setKernelSize(5);
    up->setTau(1.0);
    up->setLabmda(0.05);
    REQUIRE(up->getScale() == 2 && up->getKernelSize() == 5 && up->getLambda() == 0.05 && up->getTau() == 1.0);
    up->setOpticalFlow(std::make_shared<ZeroFlow>());
    up->setInput(std::make_shared<ExtSource>(4));
    Mat out;
    int n = 0;
    for (up->nextFrame(out); !out.empty(); up->nextFrame(out))
    {
        REQUIRE(out.rows() == 6 * 2 - 10 && out.cols() == 8 * 2 - 10 && out.channels() == 1);
        ++n;
    }
    REQUIRE(n == 4);

    // read-ahead with the built-in flow: the same number of frames out, the source pulled one frame early;
    // 16-bit frames are accepted
    struct Source16 : IFrameSource
    {
        int n, pos = 0;
        Mat frame;
        explicit Source16(int n) : n(n), frame(6, 8, 1, SRX_16U) {}
        void reset() override { pos = 0; }
        Mat nextFrame() override { return pos++ < n ? frame : Mat(); }
    };
    BTV_L1 ahead(-1);
    ahead.set("temporalAreaRadius", 1);
    ahead.setReadAhead(true);
    REQUIRE(ahead.getInt("readAhead") == 1);
    ahead.setOpticalFlow(createOptFlow_DualTVL1());
    auto src16 = std::make_shared<Source16>(6);
    ahead.setFrameSource(src16);
    REQUIRE(!ahead.nextFrame().empty());
    REQUIRE(src16->pos == 2 * 1 + 2 + 1); // warm-up reads 2r+2 frames, read-ahead one more
    int more = 1;
    while (!ahead.nextFrame().empty())
        ++more;
    REQUIRE(more == 6);

    // the reference's default flow class as a built-in: members travel with setOpticalFlow, bad values throw
    {
        BTV_L1 fbsr(-1);
        REQUIRE(dynamic_cast<Farneback*>(fbsr.getOpticalFlow().get()) != nullptr); // the default flow
        auto fb = std::make_shared<Farneback>();
        REQUIRE(fb->pyrScale == 0.5 && fb->numLevels == 5 && fb->winSize == 13 && fb->numIters == 10 &&
                fb->polyN == 5 && fb->polySigma == 1.1 && fb->flags == 0); // lib/src/optical_flow.cpp:41-50
        fb->winSize = 9;
        fbsr.set("opticalFlow", Ptr<DenseOpticalFlow>(fb));
        fbsr.set("temporalAreaRadius", 1);
        fbsr.setFrameSource(std::make_shared<Source16>(4));
        int outs = 0;
        while (!fbsr.nextFrame().empty())
            ++outs;
        REQUIRE(outs == 4);
        fb->flags = 4; // an initial flow is not built
        bool threw = false;
        try
        {
            fbsr.setOpticalFlow(fb);
        }
        catch (const Exception&)
        {
            threw = true;
        }
        REQUIRE(threw);
    }
    // flow parameters set by name before the first frame stay set (the default Farneback counts as bound, so the
    // first nextFrame() does not send the object's defaults over them), and editing another member of the flow
    // object afterwards sends that member without taking the named value back
    {
        BTV_L1 named(-1);
        named.setFlowParam("numLevels", 3);
        named.setFlowParam("polySigma", 1.5);
        named.set("temporalAreaRadius", 1);
        named.setFrameSource(std::make_shared<Source16>(4));
        REQUIRE(!named.nextFrame().empty());
        int levels = 0, win = 0;
        double sigma = 0;
        REQUIRE(srx_flow_get_int(named.handle(), "numLevels", &levels) == SRX_OK && levels == 3);
        REQUIRE(srx_flow_get_double(named.handle(), "polySigma", &sigma) == SRX_OK && sigma == 1.5);
        Farneback* fb = dynamic_cast<Farneback*>(named.getOpticalFlow().get());
        REQUIRE(fb && fb->numLevels == 3 && fb->polySigma == 1.5);
        fb->winSize = 9;
        REQUIRE(!named.nextFrame().empty());
        REQUIRE(srx_flow_get_int(named.handle(), "winSize", &win) == SRX_OK && win == 9);
        REQUIRE(srx_flow_get_int(named.handle(), "numLevels", &levels) == SRX_OK && levels == 3);
        named.setFlowParam("numIters", 4);
        REQUIRE(!named.nextFrame().empty());
        int iters = 0;
        REQUIRE(srx_flow_get_int(named.handle(), "numIters", &iters) == SRX_OK && iters == 4);
    }
    std::puts("facade ok");
    return 0;
}
