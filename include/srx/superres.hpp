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
// (samples/quality_test/main.cpp:261-267): parameters by name.
class Algorithm
{
  public:
    virtual ~Algorithm() {}
    virtual void set(const std::string& name, int v) = 0;
    virtual void set(const std::string& name, double v) = 0;
    virtual void set(const std::string& name, const Ptr<DenseOpticalFlow>& flow) = 0;
    virtual int getInt(const std::string& name) const = 0;
    virtual double getDouble(const std::string& name) const = 0;
};

// lib/include/super_resolution.hpp:146-163
class SuperResolution : public Algorithm, public IFrameSource
{
  public:
    virtual ~SuperResolution() {}
    virtual void setFrameSource(const Ptr<IFrameSource>& frameSource) = 0;
    virtual void setOpticalFlow(const Ptr<DenseOpticalFlow>& flow) = 0;
};

// SuperResolution.BTV_L1 (lib/include/super_resolution.hpp:165-195)
class BTV_L1 : public SuperResolution
{
  public:
    // device: GPU ordinal; -1 creates a dry-run object (host bookkeeping only)
    explicit BTV_L1(int device = 0)
    {
        check(srx_create_btvl1(device, &h_));
        flow_ = std::make_shared<Farneback>(); // opticalFlow = new Farneback, lib/src/btv_l1.cpp:287 (the library's default)
    }
    ~BTV_L1() override { srx_destroy(h_); }
    BTV_L1(const BTV_L1&) = delete;
    BTV_L1& operator=(const BTV_L1&) = delete;

    // ---- Algorithm::set / get with the names of lib/src/super_resolution.cpp:83-93
    void set(const std::string& name, int v) override { check(srx_set_int(h_, name.c_str(), v)); }
    void set(const std::string& name, double v) override { check(srx_set_double(h_, name.c_str(), v)); }
    void set(const std::string& name, const Ptr<DenseOpticalFlow>& flow) override
    {
        if (name != "opticalFlow")
            throw Exception(SRX_ERR_INVALID, "unknown algorithm parameter: " + name);
        setOpticalFlow(flow);
    }
    int getInt(const std::string& name) const override
    {
        int v = 0;
        check(srx_get_int(h_, name.c_str(), &v));
        return v;
    }
    double getDouble(const std::string& name) const override
    {
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

    void setOpticalFlow(const Ptr<DenseOpticalFlow>& flow) override
    {
        flow_ = flow;
        if (!flow)
            check(srx_set_flow_builtin(h_, SRX_FLOW_NONE));
        else if (dynamic_cast<DualTVL1*>(flow.get()))
            check(srx_set_flow_builtin(h_, SRX_FLOW_TVL1));
        else if (const Farneback* fb = dynamic_cast<Farneback*>(flow.get()))
        {
            check(srx_set_flow_builtin(h_, SRX_FLOW_FARNEBACK));
            setFlowParam("pyrScale", fb->pyrScale);
            setFlowParam("numLevels", fb->numLevels);
            setFlowParam("winSize", fb->winSize);
            setFlowParam("numIters", fb->numIters);
            setFlowParam("polyN", fb->polyN);
            setFlowParam("polySigma", fb->polySigma);
            setFlowParam("flags", fb->flags);
        }
        else
            check(srx_set_flow_callback(h_, &BTV_L1::flow_thunk, this));
    }
    Ptr<DenseOpticalFlow> getOpticalFlow() const { return flow_; }
    void setFlowParam(const std::string& name, int v) { check(srx_flow_set_int(h_, name.c_str(), v)); }
    void setFlowParam(const std::string& name, double v) { check(srx_flow_set_double(h_, name.c_str(), v)); }

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
        srx_image out{};
        check(srx_next_frame(h_, &out));
        return out.rows == 0 ? Mat() : Mat(out);
    }
    void nextFrame(Mat& frame) { frame = nextFrame(); } // upstream spelling
    void setOutputRange(int first, int last) { check(srx_set_output_range(h_, first, last)); }
    void setOutputOnDevice(bool on) { set("outputMemory", on ? SRX_MEM_DEVICE : SRX_MEM_HOST); }
    void setReadAhead(bool on) { set("readAhead", on ? 1 : 0); } // next frame's flow beside the current solve

    // ---- BTV_L1_Base::process (lib/src/btv_l1.cpp:297-355); dst is allocated (host) when empty
    void process(const std::vector<Mat>& src, Mat& dst, int baseIdx = 0) { process_impl(src, dst, nullptr, baseIdx); }
    void process(const std::vector<Mat>& src, Mat& dst, const std::vector<Mat>& forwardMotions, int baseIdx = 0)
    {
        if (forwardMotions.size() + 1 != src.size())
            throw Exception(SRX_ERR_INVALID, "forwardMotions must have src.size() - 1 entries");
        process_impl(src, dst, &forwardMotions, baseIdx);
    }
    // DenseOpticalFlow::calc through this object's flow (lib/include/optical_flow.hpp:44)
    void calcFlow(const Mat& frame0, const Mat& frame1, Mat& flow1, Mat* flow2 = nullptr)
    {
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
        if (flow_)
            flow_->collectGarbage();
        check(srx_collect_garbage(h_));
    }
    srx_handle* handle() const { return h_; }

  private:
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
    static int flow_thunk(void* user, const srx_image* f0, const srx_image* f1, srx_image* flow)
    {
        BTV_L1* self = static_cast<BTV_L1*>(user);
        try
        {
            self->flow_->calc(Mat(*f0), Mat(*f1), self->flow_out_, nullptr);
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
    void process_impl(const std::vector<Mat>& src, Mat& dst, const std::vector<Mat>* motions, int baseIdx)
    {
        if (src.empty())
            throw Exception(SRX_ERR_INVALID, "process: no frames");
        std::vector<srx_image> f, m;
        for (const Mat& s : src)
            f.push_back(s.view());
        if (motions)
            for (const Mat& s : *motions)
                m.push_back(s.view());
        const int s = getScale(), B = getKernelSize();
        if (dst.empty())
            dst.create(src[0].rows() * s - 2 * B, src[0].cols() * s - 2 * B, src[0].channels(), src[0].depth());
        srx_image out = dst.view();
        check(srx_process(h_, f.data(), (int)f.size(), motions ? m.data() : nullptr, baseIdx, &out));
    }

    srx_handle* h_ = nullptr;
    Ptr<IFrameSource> source_;
    Ptr<FrameSource> ext_source_;
    Ptr<DenseOpticalFlow> flow_;
    Mat cur_, flow_out_;
};

// upstream factory spelling of `new BTV_L1` (samples/video/main.cpp:106)
inline Ptr<BTV_L1> createSuperResolution_BTVL1(int device = 0)
{
    return std::make_shared<BTV_L1>(device);
}

// initModule_superres (lib/src/super_resolution.cpp:109-123): nothing to register here
inline bool initModule_superres()
{
    return true;
}

} // namespace srx
