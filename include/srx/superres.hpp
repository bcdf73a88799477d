// C++ facade over the srx C ABI (include/srx.h): the reference's classes
// without OpenCV.
//
// Spellings of the reference snapshot (lib/include/super_resolution.hpp,
// lib/include/optical_flow.hpp):
//     BTV_L1, SuperResolution::setFrameSource / reset / nextFrame,
//     IFrameSource, DenseOpticalFlow::calc / collectGarbage,
//     Algorithm-style set("scale", 4) / getInt("btvKernelSize")
// and the spellings the same code took in OpenCV's superres module:
//     createSuperResolution_BTVL1(), setInput(Ptr<FrameSource>),
//     nextFrame(Mat&), DenseOpticalFlowExt, setScale()/getScale() ...
// Both sets are thin aliases over one implementation; all arithmetic runs in
// libsrx.so on the GPU.  Header-only; link with -lsrx.
#pragma once

#include "../srx.h"

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace srx
{

template <typename T> using Ptr = std::shared_ptr<T>;

// thrown where the reference would raise cv::Exception (or, in release builds, run into undefined behaviour)
class Exception : public std::runtime_error
{
  public:
    Exception(int code, const std::string& msg) : std::runtime_error(msg), code(code) {}
    int code;
};

inline void check(int rc)
{
    if (rc != SRX_OK)
        throw Exception(rc, srx_last_error());
}

// Minimal stand-in for cv::Mat: a ref-counted or borrowed dense image in host or device memory.
inline size_t elemSize1(int depth) { return depth == SRX_8U ? 1 : (depth == SRX_16U ? 2 : 4); }

class Mat
{
  public:
    Mat() = default;
    Mat(int rows, int cols, int channels, int depth) { create(rows, cols, channels, depth); }
    // borrowed memory (host or device); the caller keeps it alive
    Mat(int rows, int cols, int channels, int depth, void* data, size_t step = 0, int mem = SRX_MEM_HOST)
    {
        im_ = srx_image{data, rows, cols, channels, depth, step, mem};
    }
    explicit Mat(const srx_image& im) : im_(im) {}

    void create(int rows, int cols, int channels, int depth)
    {
        const size_t bytes = (size_t)rows * cols * channels * elemSize1(depth);
        if (!own_ || own_->size() != bytes || own_.use_count() > 1)
            own_ = std::make_shared<std::vector<uint8_t>>(bytes);
        im_ = srx_image{own_->data(), rows, cols, channels, depth, 0, SRX_MEM_HOST};
    }
    // rows == 0 marks the empty Mat; a dry-run object hands out sized frames without data
    bool empty() const { return im_.rows == 0 || im_.cols == 0; }
    int rows() const { return im_.rows; }
    int cols() const { return im_.cols; }
    int channels() const { return im_.channels; }
    int depth() const { return im_.depth; }
    bool onDevice() const { return im_.mem == SRX_MEM_DEVICE; }
    size_t step() const { return im_.step ? im_.step : (size_t)im_.cols * im_.channels * elemSize1(im_.depth); }
    template <typename T> T* ptr(int y = 0) const { return (T*)((uint8_t*)im_.data + (size_t)y * step()); }
    const srx_image& view() const { return im_; }
    // borrowed view of a rectangle, like cv::Mat::operator()(Rect); shares this Mat's memory
    Mat roi(int x, int y, int width, int height) const
    {
        if (x < 0 || y < 0 || width < 1 || height < 1 || x + width > im_.cols || y + height > im_.rows)
            throw Exception(SRX_ERR_INVALID, "Mat::roi: rectangle outside the image");
        Mat m;
        m.own_ = own_;
        m.im_ = im_;
        m.im_.step = step();
        m.im_.data = (uint8_t*)im_.data + (size_t)y * step() + (size_t)x * im_.channels * elemSize1(im_.depth);
        m.im_.rows = height, m.im_.cols = width;
        return m;
    }
    // deep copy of a host image (the frame returned by nextFrame() aliases an internal buffer)
    Mat clone() const
    {
        Mat m;
        if (empty() || onDevice())
            return m;
        m.create(im_.rows, im_.cols, im_.channels, im_.depth);
        const size_t row = (size_t)im_.cols * im_.channels * elemSize1(im_.depth);
        for (int y = 0; y < im_.rows; ++y)
            std::memcpy(m.ptr<uint8_t>(y), ptr<uint8_t>(y), row);
        return m;
    }

  private:
    srx_image im_{};
    std::shared_ptr<std::vector<uint8_t>> own_;
};

// The two metrics of the reference's quality harness, computed on the GPU.
// cv::PSNR as called at samples/quality_test/main.cpp:320,323 (8-bit images)
inline double PSNR(const Mat& a, const Mat& b)
{
    double v = 0;
    check(srx_quality_psnr(&a.view(), &b.view(), &v, nullptr));
    return v;
}
// MSSIM() of samples/quality_test/main.cpp:140-199; returns what that function returns (for colour
// images (m0 + m1 + m3) / 3, see srx.h).  per_channel, if given, receives the 4 channel means.
inline double MSSIM(const Mat& a, const Mat& b, double* per_channel = nullptr)
{
    double v = 0;
    check(srx_quality_mssim(&a.view(), &b.view(), per_channel, &v, nullptr));
    return v;
}

// videostab::IFrameSource as the reference uses it (samples/quality_test/main.cpp:43-50)
class IFrameSource
{
  public:
    virtual ~IFrameSource() {}
    virtual void reset() = 0;
    virtual Mat nextFrame() = 0; // empty Mat = end of stream
};

class NullFrameSource : public IFrameSource
{
  public:
    void reset() override {}
    Mat nextFrame() override { return Mat(); }
};

// upstream cv::superres::FrameSource
class FrameSource
{
  public:
    virtual ~FrameSource() {}
    virtual void nextFrame(Mat& frame) = 0;
    virtual void reset() = 0;
};

// lib/include/optical_flow.hpp:41-46
class DenseOpticalFlow
{
  public:
    virtual ~DenseOpticalFlow() {}
    // flow1: rows x cols x 2 float (dx, dy interleaved); flow2 given: planar u and v
    virtual void calc(const Mat& frame0, const Mat& frame1, Mat& flow1, Mat* flow2 = nullptr) = 0;
    virtual void collectGarbage() {}
    // true: the super-resolution object asks for the two-output form calc(f0, f1, u, &v), as the reference's
    // GPU class does (lib/src/btv_l1_gpu.cpp:243,460); false: flow1 only, as lib/src/btv_l1.cpp:315,522
    virtual bool planar() const { return false; }
};
using DenseOpticalFlowExt = DenseOpticalFlow; // upstream spelling

// Marker for the GPU TV-L1 flow built into the library; parameters live on the owning BTV_L1.
class DualTVL1 : public DenseOpticalFlow
{
  public:
    void calc(const Mat&, const Mat&, Mat&, Mat*) override
    {
        throw Exception(SRX_ERR_STATE, "DualTVL1 runs inside the super-resolution object; use BTV_L1::calcFlow");
    }
};
inline Ptr<DenseOpticalFlow> createOptFlow_DualTVL1()
{
    return std::make_shared<DualTVL1>();
}

// The reference's default flow (lib/include/optical_flow.hpp class Farneback; defaults of
// lib/src/optical_flow.cpp:41-50), here the GPU implementation built into the library.  The public
// members are applied when the object is handed to setOpticalFlow().
class Farneback : public DenseOpticalFlow
{
  public:
    double pyrScale = 0.5;
    int numLevels = 5;
    int winSize = 13;
    int numIters = 10;
    int polyN = 5;
    double polySigma = 1.1;
    int flags = 0;
    void calc(const Mat&, const Mat&, Mat&, Mat*) override
    {
        throw Exception(SRX_ERR_STATE, "Farneback runs inside the super-resolution object; use BTV_L1::calcFlow");
    }
};
inline Ptr<DenseOpticalFlow> createOptFlow_Farneback()
{
    return std::make_shared<Farneback>();
}

// The part of cv::Algorithm that the reference's programs use on a Ptr<SuperResolution>
// (samples/quality_test/main.cpp:261-267): parameters by name, and creation by registered name
// (Algorithm::create<SuperResolution>("SuperResolution.BTV_L1") once initModule_superres() has run).
class Algorithm
{
  public:
    virtual ~Algorithm() {}
    virtual void set(const std::string& name, int v) = 0;
    virtual void set(const std::string& name, double v) = 0;
    virtual void set(const std::string& name, const Ptr<DenseOpticalFlow>& flow) = 0;
    virtual int getInt(const std::string& name) const = 0;
    virtual double getDouble(const std::string& name) const = 0;
    virtual std::string name() const { return std::string(); }

    typedef Ptr<Algorithm> (*Constructor)(int device);
    // registered names, in registration order (cv::Algorithm::getList)
    static void getList(std::vector<std::string>& names)
    {
        names.clear();
        for (const auto& e : registry())
            names.push_back(e.first);
    }
    // cv::Algorithm::create<T>(name): an empty Ptr for a name that is not registered or is of another type
    template <typename T> static Ptr<T> create(const std::string& name)
    {
        for (const auto& e : registry())
            if (e.first == name)
                return std::dynamic_pointer_cast<T>(e.second(defaultDevice()));
        return Ptr<T>();
    }
    // objects made by create() live on this GPU (-1: dry-run objects, host bookkeeping only)
    static int& defaultDevice()
    {
        static int device = 0;
        return device;
    }
    static bool registerAlgorithm(const std::string& name, Constructor make)
    {
        for (const auto& e : registry())
            if (e.first == name)
                return true;
        registry().emplace_back(name, make);
        return true;
    }

  private:
    static std::vector<std::pair<std::string, Constructor>>& registry()
    {
        static std::vector<std::pair<std::string, Constructor>> r;
        return r;
    }
};

// lib/include/super_resolution.hpp:46-88.  The parameters are public members, as in the reference; they are
// handed to the library at every process() call, where out-of-range values (which the reference only
// debug-asserts, lib/src/btv_l1.cpp:359-365) raise srx::Exception.
class BTV_L1_Base
{
  public:
    // device: GPU ordinal; -1 creates a dry-run object (host bookkeeping only)
    explicit BTV_L1_Base(int device = 0)
    {
        check(srx_create_btvl1(device, &h_));
        opticalFlow = std::make_shared<Farneback>(); // opticalFlow = new Farneback, lib/src/btv_l1.cpp:287
        // A new library object already runs Farneback with the defaults that fb_sent_ holds, so the default flow
        // counts as bound: the first pushParams() sends nothing and leaves earlier setFlowParam() values alone.
        bound_flow_ = opticalFlow;
    }
    virtual ~BTV_L1_Base() { srx_destroy(h_); }
    BTV_L1_Base(const BTV_L1_Base&) = delete;
    BTV_L1_Base& operator=(const BTV_L1_Base&) = delete;

    // lib/src/btv_l1.cpp:297-355; dst is allocated (host) when empty.  Without forwardMotions the motions come
    // from opticalFlow (lib/src/btv_l1.cpp:297-321).
    void process(const std::vector<Mat>& src, Mat& dst, int baseIdx = 0) { process_impl(src, dst, nullptr, baseIdx); }
    void process(const std::vector<Mat>& src, Mat& dst, const std::vector<Mat>& forwardMotions, int baseIdx = 0)
    {
        if (forwardMotions.size() + 1 != src.size())
            throw Exception(SRX_ERR_INVALID, "forwardMotions must have src.size() - 1 entries");
        process_impl(src, dst, &forwardMotions, baseIdx);
    }

    // lib/include/super_resolution.hpp:54-62, defaults of lib/src/btv_l1.cpp:277-288
    int scale = 4;
    int iterations = 180;
    double tau = 1.3;
    double lambda = 0.03;
    double alpha = 0.7;
    int btvKernelSize = 7;
    int blurKernelSize = 5;
    double blurSigma = 0.0;
    Ptr<DenseOpticalFlow> opticalFlow;

    // The C handle.  The public members above are the parameters' home: every process() / nextFrame() / calcFlow()
    // sends members that differ from the library's values, so a solver parameter set through the handle alone
    // lasts until the next such call; set the member (or BTV_L1::set) instead.
    srx_handle* handle() const { return h_; }

  protected:
    // members -> library.  Values the library already holds are not sent again.
    void pushParams()
    {
        pushInt("scale", scale);
        pushInt("iterations", iterations);
        pushDouble("tau", tau);
        pushDouble("lambda", lambda);
        pushDouble("alpha", alpha);
        pushInt("btvKernelSize", btvKernelSize);
        pushInt("blurKernelSize", blurKernelSize);
        pushDouble("blurSigma", blurSigma);
        if (opticalFlow != bound_flow_ || flowChanged())
            bindFlow();
    }
    void pushInt(const char* name, int v)
    {
        int cur = 0;
        check(srx_get_int(h_, name, &cur));
        if (cur != v)
            check(srx_set_int(h_, name, v));
    }
    void pushDouble(const char* name, double v)
    {
        double cur = 0;
        check(srx_get_double(h_, name, &cur));
        if (cur != v)
            check(srx_set_double(h_, name, v));
    }
    // the parameters of a built-in Farneback object are public members too: resend when they were edited
    bool flowChanged() const
    {
        const Farneback* fb = dynamic_cast<const Farneback*>(opticalFlow.get());
        return fb && (fb->pyrScale != fb_sent_.pyrScale || fb->numLevels != fb_sent_.numLevels ||
                      fb->winSize != fb_sent_.winSize || fb->numIters != fb_sent_.numIters ||
                      fb->polyN != fb_sent_.polyN || fb->polySigma != fb_sent_.polySigma ||
                      fb->flags != fb_sent_.flags);
    }
    void bindFlow()
    {
        DenseOpticalFlow* f = opticalFlow.get();
        if (!f)
            check(srx_set_flow_builtin(h_, SRX_FLOW_NONE));
        else if (dynamic_cast<DualTVL1*>(f))
            check(srx_set_flow_builtin(h_, SRX_FLOW_TVL1));
        else if (const Farneback* fb = dynamic_cast<Farneback*>(f))
        {
            check(srx_set_flow_builtin(h_, SRX_FLOW_FARNEBACK));
            check(srx_flow_set_double(h_, "pyrScale", fb->pyrScale));
            check(srx_flow_set_int(h_, "numLevels", fb->numLevels));
            check(srx_flow_set_int(h_, "winSize", fb->winSize));
            check(srx_flow_set_int(h_, "numIters", fb->numIters));
            check(srx_flow_set_int(h_, "polyN", fb->polyN));
            check(srx_flow_set_double(h_, "polySigma", fb->polySigma));
            check(srx_flow_set_int(h_, "flags", fb->flags));
            fb_sent_.pyrScale = fb->pyrScale, fb_sent_.numLevels = fb->numLevels, fb_sent_.winSize = fb->winSize;
            fb_sent_.numIters = fb->numIters, fb_sent_.polyN = fb->polyN, fb_sent_.polySigma = fb->polySigma;
            fb_sent_.flags = fb->flags;
        }
        else if (f->planar())
            check(srx_set_flow_callback2(h_, &BTV_L1_Base::flow2_thunk, this));
        else
            check(srx_set_flow_callback(h_, &BTV_L1_Base::flow_thunk, this));
        bound_flow_ = opticalFlow;
    }

    void mirrorFlowParam(const std::string& name, double v)
    {
        Farneback* fb = dynamic_cast<Farneback*>(opticalFlow.get());
        if (!fb || opticalFlow != bound_flow_)
            return;
        for (Farneback* o : {fb, &fb_sent_})
        {
            if (name == "pyrScale") o->pyrScale = v;
            else if (name == "numLevels") o->numLevels = (int)v;
            else if (name == "winSize") o->winSize = (int)v;
            else if (name == "numIters") o->numIters = (int)v;
            else if (name == "polyN") o->polyN = (int)v;
            else if (name == "polySigma") o->polySigma = v;
            else if (name == "flags") o->flags = (int)v;
        }
    }

    srx_handle* h_ = nullptr;

  private:
    // The results stay alive in flow_out_ / flow_out2_ until the plug-in is called again, which is as long as
    // the library may read them.
    static int flow_thunk(void* user, const srx_image* f0, const srx_image* f1, srx_image* flow)
    {
        BTV_L1_Base* self = static_cast<BTV_L1_Base*>(user);
        try
        {
            self->bound_flow_->calc(Mat(*f0), Mat(*f1), self->flow_out_, nullptr);
        }
        catch (...)
        {
            return 1;
        }
        if (self->flow_out_.empty())
            return 1;
        *flow = self->flow_out_.view();
        return 0;
    }
    // calc(frame0, frame1, flow1, flow2): planar u and v (lib/src/optical_flow.cpp:71-81)
    static int flow2_thunk(void* user, const srx_image* f0, const srx_image* f1, srx_image* u, srx_image* v)
    {
        BTV_L1_Base* self = static_cast<BTV_L1_Base*>(user);
        try
        {
            self->bound_flow_->calc(Mat(*f0), Mat(*f1), self->flow_out_, &self->flow_out2_);
        }
        catch (...)
        {
            return 1;
        }
        if (self->flow_out_.empty() || self->flow_out2_.empty())
            return 1;
        *u = self->flow_out_.view();
        *v = self->flow_out2_.view();
        return 0;
    }
    void process_impl(const std::vector<Mat>& src, Mat& dst, const std::vector<Mat>* motions, int baseIdx)
    {
        if (src.empty())
            throw Exception(SRX_ERR_INVALID, "process: no frames");
        pushParams();
        std::vector<srx_image> f, m;
        for (const Mat& s : src)
            f.push_back(s.view());
        if (motions)
            for (const Mat& s : *motions)
                m.push_back(s.view());
        if (dst.empty())
            dst.create(src[0].rows() * scale - 2 * btvKernelSize, src[0].cols() * scale - 2 * btvKernelSize,
                       src[0].channels(), src[0].depth());
        srx_image out = dst.view();
        check(srx_process(h_, f.data(), (int)f.size(), motions ? m.data() : nullptr, baseIdx, &out));
    }

    Ptr<DenseOpticalFlow> bound_flow_; // what the library's call-back reaches
    Farneback fb_sent_;
    Mat flow_out_, flow_out2_;
};

// lib/include/super_resolution.hpp:146-163
class SuperResolution : public Algorithm, public IFrameSource
{
  public:
    virtual ~SuperResolution() {}
    virtual void setFrameSource(const Ptr<IFrameSource>& frameSource) = 0;
    virtual void setOpticalFlow(const Ptr<DenseOpticalFlow>& flow) = 0;
};

// SuperResolution.BTV_L1 (lib/include/super_resolution.hpp:165-195).  The reference derives privately from
// BTV_L1_Base and reaches the solver's parameters by name only; here the process() overloads stay reachable
// as well (programs of this repository use them on the streaming object).
class BTV_L1 : public SuperResolution, private BTV_L1_Base
{
  public:
    explicit BTV_L1(int device = 0) : BTV_L1_Base(device) {}
    BTV_L1(const BTV_L1&) = delete;
    BTV_L1& operator=(const BTV_L1&) = delete;

    int temporalAreaRadius = 4; // lib/include/super_resolution.hpp:172; read when the stream (re)starts

    std::string name() const override { return "SuperResolution.BTV_L1"; }
    using BTV_L1_Base::handle;
    using BTV_L1_Base::process;

    // ---- Algorithm::set / get with the names of lib/src/super_resolution.cpp:83-93.  A value the library
    // refuses raises srx::Exception and leaves the parameter as it was.
    void set(const std::string& name, int v) override
    {
        check(srx_set_int(h_, name.c_str(), v));
        if (int* m = intMember(name))
            *m = v;
    }
    void set(const std::string& name, double v) override
    {
        check(srx_set_double(h_, name.c_str(), v));
        if (double* m = doubleMember(name))
            *m = v;
    }
    void set(const std::string& name, const Ptr<DenseOpticalFlow>& flow) override
    {
        if (name != "opticalFlow")
            throw Exception(SRX_ERR_INVALID, "unknown algorithm parameter: " + name);
        setOpticalFlow(flow);
    }
    int getInt(const std::string& name) const override
    {
        if (const int* m = const_cast<BTV_L1*>(this)->intMember(name))
            return *m;
        int v = 0;
        check(srx_get_int(h_, name.c_str(), &v));
        return v;
    }
    double getDouble(const std::string& name) const override
    {
        if (const double* m = const_cast<BTV_L1*>(this)->doubleMember(name))
            return *m;
        double v = 0;
        check(srx_get_double(h_, name.c_str(), &v));
        return v;
    }

    // ---- upstream-style accessors
    int getScale() const { return getInt("scale"); }
    void setScale(int v) { set("scale", v); }
    int getIterations() const { return getInt("iterations"); }
    void setIterations(int v) { set("iterations", v); }
    double getTau() const { return getDouble("tau"); }
    void setTau(double v) { set("tau", v); }
    double getLabmda() const { return getDouble("lambda"); } // sic: upstream's spelling
    void setLabmda(double v) { set("lambda", v); }
    double getLambda() const { return getDouble("lambda"); }
    void setLambda(double v) { set("lambda", v); }
    double getAlpha() const { return getDouble("alpha"); }
    void setAlpha(double v) { set("alpha", v); }
    int getKernelSize() const { return getInt("btvKernelSize"); }
    void setKernelSize(int v) { set("btvKernelSize", v); }
    int getBlurKernelSize() const { return getInt("blurKernelSize"); }
    void setBlurKernelSize(int v) { set("blurKernelSize", v); }
    double getBlurSigma() const { return getDouble("blurSigma"); }
    void setBlurSigma(double v) { set("blurSigma", v); }
    int getTemporalAreaRadius() const { return getInt("temporalAreaRadius"); }
    void setTemporalAreaRadius(int v) { set("temporalAreaRadius", v); }

    // A refused flow object (e.g. Farneback flags the library does not build) raises srx::Exception.
    void setOpticalFlow(const Ptr<DenseOpticalFlow>& flow) override
    {
        opticalFlow = flow;
        bindFlow();
    }
    Ptr<DenseOpticalFlow> getOpticalFlow() const { return opticalFlow; }
    // Parameters of the built-in flows by name.  Where the flow object is a Farneback, whose parameters are public
    // members, the member takes the value too, so that a later edit of another member does not send the old one.
    void setFlowParam(const std::string& name, int v)
    {
        check(srx_flow_set_int(h_, name.c_str(), v));
        mirrorFlowParam(name, (double)v);
    }
    void setFlowParam(const std::string& name, double v)
    {
        check(srx_flow_set_double(h_, name.c_str(), v));
        mirrorFlowParam(name, v);
    }

    // ---- SuperResolution (lib/src/super_resolution.cpp:131-152)
    void setFrameSource(const Ptr<IFrameSource>& frameSource) override
    {
        source_ = frameSource;
        ext_source_.reset();
        check(srx_set_input(h_, &BTV_L1::next_thunk, &BTV_L1::reset_thunk, this));
    }
    void setInput(const Ptr<FrameSource>& frameSource) // upstream spelling
    {
        ext_source_ = frameSource;
        source_.reset();
        check(srx_set_input(h_, &BTV_L1::next_thunk, &BTV_L1::reset_thunk, this));
    }
    void reset() override { check(srx_reset(h_)); }
    // The returned Mat aliases a buffer of this object, valid until the next call (as in the reference,
    // lib/src/btv_l1.cpp:504-505); clone() to keep it.
    Mat nextFrame() override
    {
        pushParams();
        pushInt("temporalAreaRadius", temporalAreaRadius);
        srx_image out{};
        check(srx_next_frame(h_, &out));
        return out.rows == 0 ? Mat() : Mat(out);
    }
    void nextFrame(Mat& frame) { frame = nextFrame(); } // upstream spelling
    void setOutputRange(int first, int last) { check(srx_set_output_range(h_, first, last)); }
    void setOutputOnDevice(bool on) { set("outputMemory", on ? SRX_MEM_DEVICE : SRX_MEM_HOST); }
    void setReadAhead(bool on) { set("readAhead", on ? 1 : 0); } // next frame's flow beside the current solve
    void setFlowPairsInFlight(int n) { set("flowPairsInFlight", n); } // process(): a window's TV-L1 flows side by side

    // DenseOpticalFlow::calc through this object's flow (lib/include/optical_flow.hpp:44)
    void calcFlow(const Mat& frame0, const Mat& frame1, Mat& flow1, Mat* flow2 = nullptr)
    {
        pushParams();
        if (flow2)
        {
            flow1.create(frame0.rows(), frame0.cols(), 1, SRX_32F);
            flow2->create(frame0.rows(), frame0.cols(), 1, SRX_32F);
            srx_image a = flow1.view(), b = flow2->view();
            check(srx_flow_calc(h_, &frame0.view(), &frame1.view(), &a, &b));
        }
        else
        {
            flow1.create(frame0.rows(), frame0.cols(), 2, SRX_32F);
            srx_image a = flow1.view();
            check(srx_flow_calc(h_, &frame0.view(), &frame1.view(), &a, nullptr));
        }
    }
    void collectGarbage()
    {
        if (opticalFlow)
            opticalFlow->collectGarbage();
        check(srx_collect_garbage(h_));
    }

  private:
    int* intMember(const std::string& n)
    {
        return n == "scale" ? &scale : n == "iterations" ? &iterations : n == "btvKernelSize" ? &btvKernelSize
             : n == "blurKernelSize" ? &blurKernelSize : n == "temporalAreaRadius" ? &temporalAreaRadius : nullptr;
    }
    double* doubleMember(const std::string& n)
    {
        return n == "tau" ? &tau : n == "lambda" ? &lambda : n == "alpha" ? &alpha : n == "blurSigma" ? &blurSigma
                                                                                                        : nullptr;
    }
    static int next_thunk(void* user, srx_image* out)
    {
        BTV_L1* self = static_cast<BTV_L1*>(user);
        try
        {
            if (self->source_)
                self->cur_ = self->source_->nextFrame();
            else if (self->ext_source_)
                self->ext_source_->nextFrame(self->cur_);
            else
                self->cur_ = Mat();
        }
        catch (...)
        {
            self->cur_ = Mat();
        }
        if (self->cur_.empty())
            return 0;
        *out = self->cur_.view();
        return 1;
    }
    static void reset_thunk(void* user)
    {
        BTV_L1* self = static_cast<BTV_L1*>(user);
        if (self->source_)
            self->source_->reset();
        else if (self->ext_source_)
            self->ext_source_->reset();
    }

    Ptr<IFrameSource> source_;
    Ptr<FrameSource> ext_source_;
    Mat cur_;
};
using BTV_L1_GPU = BTV_L1; // the reference's second class of the same interface (lib/include/super_resolution.hpp:197)

// upstream factory spelling of `new BTV_L1` (samples/video/main.cpp:106)
inline Ptr<BTV_L1> createSuperResolution_BTVL1(int device = 0)
{
    return std::make_shared<BTV_L1>(device);
}

// initModule_superres (lib/src/super_resolution.cpp:109-123): registers the algorithms this library builds under
// the reference's names, so that Algorithm::create<SuperResolution>("SuperResolution.BTV_L1") works as it does
// after the reference's call.  True when every name is in the list afterwards.  The flow classes the reference
// registers beside Farneback (Simple, Brox_GPU, PyrLK_GPU) are not built and not registered.
inline bool initModule_superres()
{
    Algorithm::registerAlgorithm("SuperResolution.BTV_L1",
                                 [](int device) -> Ptr<Algorithm> { return std::make_shared<BTV_L1>(device); });
    Algorithm::registerAlgorithm("SuperResolution.BTV_L1_GPU",
                                 [](int device) -> Ptr<Algorithm> { return std::make_shared<BTV_L1>(device); });
    std::vector<std::string> names;
    Algorithm::getList(names);
    bool btv = false, btv_gpu = false;
    for (const std::string& n : names)
        btv |= n == "SuperResolution.BTV_L1", btv_gpu |= n == "SuperResolution.BTV_L1_GPU";
    return btv && btv_gpu;
}

} // namespace srx
