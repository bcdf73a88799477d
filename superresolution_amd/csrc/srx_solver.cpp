// Host helpers and the BTV-L1 solver (host side of BTV_L1_Base::run).
#include "srx_host.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace srx
{

namespace
{
thread_local std::string g_error;
}

int g_kernel_path = 0;

void set_error(const std::string& msg)
{
    g_error = msg;
}
const std::string& last_error()
{
    return g_error;
}
int hip_fail(hipError_t e, const char* what)
{
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return SRX_ERR_DEVICE;
}

// ---------------------------------------------------------------------------
int DeviceBuffer::reserve(size_t bytes)
{
    if (bytes <= cap_)
        return 0;
    release();
    SRX_HIP(hipMalloc(&ptr_, bytes));
    cap_ = bytes;
    return 0;
}
void DeviceBuffer::release()
{
    if (ptr_)
        (void)hipFree(ptr_);
    ptr_ = nullptr;
    cap_ = 0;
}
int PinnedBuffer::reserve(size_t bytes)
{
    if (bytes <= cap_)
        return 0;
    release();
    SRX_HIP(hipHostMalloc(&ptr_, bytes, hipHostMallocDefault));
    cap_ = bytes;
    return 0;
}
void PinnedBuffer::release()
{
    if (ptr_)
        (void)hipHostFree(ptr_);
    ptr_ = nullptr;
    cap_ = 0;
}

// ---------------------------------------------------------------------------
int validate_params(const Params& p)
{
    auto bad = [](const char* m) {
        set_error(m);
        return (int)SRX_ERR_INVALID;
    };
    if (p.scale < 2 || p.scale > 16)
        return bad("scale must be in 2..16");
    if (p.iterations < 1)
        return bad("iterations must be positive");
    if (!(p.tau > 0.0))
        return bad("tau must be positive");
    if (!(p.alpha > 0.0))
        return bad("alpha must be positive");
    if (p.btvKernelSize < 1 || (p.btvKernelSize - 1) / 2 > kMaxBtvRadius)
        return bad("btvKernelSize must be in 1..16");
    if (p.blurKernelSize < 1 || p.blurKernelSize > kMaxGauss || (p.blurKernelSize & 1) == 0)
        return bad("blurKernelSize must be odd and in 1..31");
    if (!(p.blurSigma >= 0.0))
        return bad("blurSigma must be non-negative");
    if (p.temporalAreaRadius < 0 || 2 * p.temporalAreaRadius + 1 > kMaxWindow)
        return bad("temporalAreaRadius must be in 0..19");
    return SRX_OK;
}

// interpolateCubic of OpenCV's imgwarp.cpp (A = -0.75), float arithmetic
static void cubic_coeffs(float x, float* c)
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

// cv::resize tap set-up.  Columns whose tap 1 would fall outside [0, cols-1)
// are snapped to a plain copy, as OpenCV 2.4 does; rows are only clipped.
void cubic_table_x(int scols, int dcols, std::vector<int>& sx, std::vector<float>& alpha)
{
    sx.resize(dcols);
    alpha.resize((size_t)dcols * 4);
    const double scale = 1.0 / ((double)dcols / scols);
    for (int dx = 0; dx < dcols; ++dx)
    {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        int s = (int)std::floor((double)fx);
        fx -= s;
        if (s < 0)
            fx = 0, s = 0;
        if (s >= scols - 1)
            fx = 0, s = scols - 1;
        sx[dx] = s;
        cubic_coeffs(fx, &alpha[(size_t)dx * 4]);
    }
}

void cubic_table_y(int srows, int drows, std::vector<int>& sy, std::vector<float>& beta)
{
    sy.resize(drows);
    beta.resize((size_t)drows * 4);
    const double scale = 1.0 / ((double)drows / srows);
    for (int dy = 0; dy < drows; ++dy)
    {
        float fy = (float)((dy + 0.5) * scale - 0.5);
        const int s = (int)std::floor((double)fy);
        fy -= s;
        sy[dy] = s;
        cubic_coeffs(fy, &beta[(size_t)dy * 4]);
    }
}

// cv::getGaussianKernel(n, sigma, CV_32F)
int gaussian_kernel(int n, double sigma, float* k)
{
    if (n < 1 || n > kMaxGauss || (n & 1) == 0)
    {
        set_error("gaussian kernel size must be odd and in 1..31");
        return SRX_ERR_INVALID;
    }
    static const float tab[4][7] = {{1.f},
                                    {0.25f, 0.5f, 0.25f},
                                    {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
                                    {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    const float* fixed = (n <= 7 && sigma <= 0) ? tab[n >> 1] : nullptr;
    const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; ++i)
    {
        const double x = i - (n - 1) * 0.5;
        const double t = fixed ? (double)fixed[i] : std::exp(scale2X * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i)
        k[i] = (float)(k[i] * sum);
    return SRX_OK;
}

// calcBtvWeights (lib/src/btv_l1.cpp:172-189)
int calc_btv_weights(int btv_kernel_size, double alpha, float* w)
{
    const int ksize = (btv_kernel_size - 1) / 2;
    const float alpha_f = (float)alpha;
    int ind = 0;
    for (int m = 0; m <= ksize; ++m)
        for (int l = ksize; l + m >= 0; --l, ++ind)
            w[ind] = (float)std::pow((double)alpha_f, (double)(std::abs(m) + std::abs(l)));
    return ind;
}

// ---------------------------------------------------------------------------
int CubicTablesDev::ensure(int srows, int scols, int drows, int dcols, hipStream_t st)
{
    if (srows == srows_ && scols == scols_ && drows == drows_ && dcols == dcols_)
        return 0;
    std::vector<int> sx, sy;
    std::vector<float> al, be;
    cubic_table_x(scols, dcols, sx, al);
    cubic_table_y(srows, drows, sy, be);
    // the previous tables may still be in use by queued kernels
    SRX_HIP(hipStreamSynchronize(st));
    if (int rc = xofs_.reserve(sx.size() * sizeof(int)))
        return rc;
    if (int rc = alpha_.reserve(al.size() * sizeof(float)))
        return rc;
    if (int rc = yofs_.reserve(sy.size() * sizeof(int)))
        return rc;
    if (int rc = beta_.reserve(be.size() * sizeof(float)))
        return rc;
    SRX_HIP(hipMemcpyAsync(xofs_.get(), sx.data(), sx.size() * sizeof(int), hipMemcpyHostToDevice, st));
    SRX_HIP(hipMemcpyAsync(alpha_.get(), al.data(), al.size() * sizeof(float), hipMemcpyHostToDevice, st));
    SRX_HIP(hipMemcpyAsync(yofs_.get(), sy.data(), sy.size() * sizeof(int), hipMemcpyHostToDevice, st));
    SRX_HIP(hipMemcpyAsync(beta_.get(), be.data(), be.size() * sizeof(float), hipMemcpyHostToDevice, st));
    SRX_HIP(hipStreamSynchronize(st)); // the host vectors go out of scope
    srows_ = srows, scols_ = scols, drows_ = drows, dcols_ = dcols;
    return 0;
}

CubicTables CubicTablesDev::view() const
{
    return CubicTables{xofs_.as<int>(), alpha_.as<float4>(), yofs_.as<int>(), beta_.as<float4>()};
}

// ---------------------------------------------------------------------------
int Solver::ensure(int K, int rows, int cols, int ch, int scale, hipStream_t st)
{
    const size_t hr_px = (size_t)rows * scale * cols * scale;
    // HR images carry one extra zero pixel; sign buffers are sized for the pair plane of the
    // specialised kernels (the generic ones use the first rows*cols words)
    const size_t x_bytes = plane_stride_for(hr_px) * ch * sizeof(float);
    const size_t sign_bytes = sign_pair_bytes(rows, cols);
    const size_t rel_bytes = (size_t)rows * cols * 2 * sizeof(float);
    const bool dec = (scale == 4 || scale == 2) && g_kernel_path == 0;
    const size_t dec_fwd_bytes = dec_fwd_words(rows * scale, cols * scale, scale) * 4;
    if ((int)fwd_maps_.size() < K)
    {
        fwd_maps_.resize(K);
        bwd_maps_.resize(K);
        dec_fwd_.resize(K);
        dec_bwd_.resize(K);
        signs_.resize(K);
        rel_.resize(K);
    }
    bool grow = x_[0].capacity() < x_bytes;
    for (int k = 0; k < K && !grow; ++k)
        grow = fwd_maps_[k].capacity() < hr_px * 4 || signs_[k].capacity() < sign_bytes ||
               rel_[k].capacity() < rel_bytes || (dec && dec_fwd_[k].capacity() < dec_fwd_bytes);
    if (grow)
        SRX_HIP(hipStreamSynchronize(st)); // buffers about to be freed may still be in use
    for (int k = 0; k < K; ++k)
    {
        if (int rc = fwd_maps_[k].reserve(hr_px * 4))
            return rc;
        if (int rc = bwd_maps_[k].reserve(hr_px * 4))
            return rc;
        if (dec)
        {
            if (int rc = dec_fwd_[k].reserve(dec_fwd_bytes))
                return rc;
            if (scale == 4)
                if (int rc = dec_bwd_[k].reserve(dec_bwd_words(hr_px) * 4))
                    return rc;
        }
        if (signs_[k].capacity() < sign_bytes)
        {
            if (int rc = signs_[k].reserve(sign_bytes))
                return rc;
            SRX_HIP(hipMemsetAsync(signs_[k].get(), 0, sign_bytes, st));
        }
        if (int rc = rel_[k].reserve(rel_bytes))
            return rc;
    }
    for (int i = 0; i < 2; ++i)
    {
        if (int rc = x_[i].reserve(x_bytes))
            return rc;
        // the zero pixel at index hr_px of every plane (planar layout of the specialised kernels)
        for (int c = 0; c < ch; ++c)
            SRX_HIP(hipMemsetAsync(x_[i].as<float>() + c * plane_stride_for(hr_px) + hr_px, 0, 4 * sizeof(float), st));
    }
    return tables_.ensure(rows, cols, rows * scale, cols * scale, st);
}

int Solver::collect_timing(hipStream_t st)
{
    if (pending_iters_ == 0)
        return SRX_OK;
    SRX_HIP(hipStreamSynchronize(st));
    for (int it = 0; it < pending_iters_; ++it)
    {
        float a = 0.f, b = 0.f;
        SRX_HIP(hipEventElapsedTime(&a, events_[4 * it], events_[4 * it + 1]));
        SRX_HIP(hipEventElapsedTime(&b, events_[4 * it + 2], events_[4 * it + 3]));
        fwd_ms_ += a;
        bwd_ms_ += b;
    }
    timed_iters_ += pending_iters_;
    pending_iters_ = 0;
    return SRX_OK;
}

void Solver::release()
{
    for (hipEvent_t e : events_)
        (void)hipEventDestroy(e);
    events_.clear();
    pending_iters_ = 0;
    fwd_maps_.clear();
    bwd_maps_.clear();
    dec_fwd_.clear();
    dec_bwd_.clear();
    signs_.clear();
    rel_.clear();
    x_[0].release();
    x_[1].release();
    tile_plan_.release();
    tile_cost_.release();
    tables_ = CubicTablesDev();
}

int Solver::run(const float* const* y, const float* const* rel, int K, int base_idx, int rows, int cols, int ch,
                const Params& p, float* out, float* x_full, hipStream_t st)
{
    const int announced_bytes = y8_count_; // byte_frames() holds for one call
    y8_count_ = 0;
    const int still = still_frame_; // set by process() for the call it makes, never by callers of run()
    still_frame_ = -1;
    if (int rc = validate_params(p))
        return rc;
    if (K < 1 || K > kMaxWindow || base_idx < 0 || base_idx >= K)
    {
        set_error("run: window size or base index out of range");
        return SRX_ERR_INVALID;
    }
    if (ch != 1 && ch != 3)
    {
        set_error("run: frames must have 1 or 3 channels");
        return SRX_ERR_INVALID;
    }
    const int s = p.scale, B = p.btvKernelSize;
    const long R = (long)rows * s, C = (long)cols * s;
    if (rows < 1 || cols < 1 || R > 32767 || C > 32767)
    {
        set_error("run: high-resolution size must be within 1..32767 per side");
        return SRX_ERR_INVALID;
    }
    if (R - 2 * B < 1 || C - 2 * B < 1)
    {
        set_error("run: image smaller than the btvKernelSize crop");
        return SRX_ERR_INVALID;
    }
    if (int rc = ensure(K, rows, cols, ch, s, st))
        return rc;

    // blur filter and btv weights are re-derived on parameter change (lib/src/btv_l1.cpp:392-405)
    if (p.blurKernelSize != curBlurKernelSize_ || p.blurSigma != curBlurSigma_)
    {
        if (int rc = gaussian_kernel(p.blurKernelSize, p.blurSigma, gauss_.k))
            return rc;
        gauss_.n = p.blurKernelSize;
        curBlurKernelSize_ = p.blurKernelSize;
        curBlurSigma_ = p.blurSigma;
    }
    if (p.btvKernelSize != curBtvKernelSize_ || p.alpha != curAlpha_)
    {
        std::memset(btv_.w, 0, sizeof(btv_.w));
        calc_btv_weights(p.btvKernelSize, p.alpha, btv_.w);
        btv_.radius = (p.btvKernelSize - 1) / 2;
        curBtvKernelSize_ = p.btvKernelSize;
        curAlpha_ = p.alpha;
    }

    const CubicTables t = tables_.view();
    // The specialised update kernel keeps X planar.  At scale 4 it reads the pair planes that only the
    // specialised forward kernel writes, so there the two are used together or not at all; at scale 2 it
    // reads plain planes, which the generic forward kernel writes (from the planar X).
    const bool fastB = g_kernel_path == 0 && backproject_fast_eligible((int)R, (int)C, ch, s, gauss_, btv_) &&
                       (s != 4 || forward_fast_eligible((int)R, (int)C, ch, s, gauss_));
    const bool fastA = fastB && s == 4;
    ps_ = fastB ? plane_stride_for((size_t)R * C) : 0;
    PtrList fwd{}, bwd{}, dfwd{}, dbwd{}, ys{}, ys8{}, sg{};
    MutPtrList sgm{};
    // 8-bit originals of the frames, announced for this call only (byte_frames)
    const bool have_bytes = (fastA || (fastB && s == 2)) && announced_bytes == K;
    for (int k = 0; k < K; ++k)
    {
        if (int rc = launch_build_maps(rel[k], rows, cols, s, t, fwd_maps_[k].as<int16_t>(),
                                       bwd_maps_[k].as<int16_t>(), fastB ? dec_fwd_[k].as<uint32_t>() : nullptr,
                                       fastA ? dec_bwd_[k].as<uint32_t>() : nullptr, st))
            return rc;
        fwd.p[k] = fwd_maps_[k].get();
        bwd.p[k] = bwd_maps_[k].get();
        dfwd.p[k] = dec_fwd_[k].get();
        dbwd.p[k] = dec_bwd_[k].get();
        ys.p[k] = y[k];
        ys8.p[k] = have_bytes ? y8_[k] : nullptr;
        sg.p[k] = signs_[k].get();
        sgm.p[k] = signs_[k].get();
    }

    // the forward kernel's tiles, cut by what the rows of this window's maps cost
    const void* tile_plan = nullptr;
    if (fastA)
    {
        size_t scratch_bytes = 0;
        const size_t plan_bytes = forward_plan_bytes((int)R, (int)C, ch, K, &scratch_bytes);
        if (int rc = tile_plan_.reserve(plan_bytes))
            return rc;
        if (int rc = tile_cost_.reserve(scratch_bytes))
            return rc;
        bool planned = false;
        if (int rc = launch_plan_forward_tiles(dbwd, K, (int)R, (int)C, ch, tile_plan_.get(), tile_cost_.get(), &planned, st))
            return rc;
        tile_plan = planned ? tile_plan_.get() : nullptr;
    }

    // initial estimation (lib/src/btv_l1.cpp:412)
    cur_ = 0;
    if (int rc = launch_resize_cubic(y[base_idx], rows, cols, ch, t, x_[0].as<float>(), (int)R, (int)C, ps_, st))
        return rc;

    if (timing_)
    {
        if (int rc = collect_timing(st)) // events of the previous run are about to be re-recorded
            return rc;
        while ((int)events_.size() < 4 * ((p.iterations + kTimingStride - 1) / kTimingStride))
        {
            hipEvent_t e;
            SRX_HIP(hipEventCreate(&e));
            events_.push_back(e);
        }
    }
    // Every kTimingStride-th iteration is timed: its two launches carry a start and a stop event each, tied to the
    // kernels' own begin and end (g_launch_timer); the averages are over those iterations.
    int timed = 0;
    struct Disarm
    {
        ~Disarm() { g_launch_timer = LaunchTimer{}; }
    } disarm;
    for (int it = 0; it < p.iterations; ++it)
    {
        const bool time_it = timing_ && it % kTimingStride == 0;
        if (time_it)
            g_launch_timer = LaunchTimer{events_[4 * timed], events_[4 * timed + 1]};
        if (int rc = launch_forward_sign(x_[cur_].as<float>(), (int)R, (int)C, ch, bwd, fastA ? &dbwd : nullptr, ys,
                                         sgm, K, s, gauss_, ps_, st, have_bytes ? &ys8 : nullptr, still, tile_plan))
            return rc;
        if (time_it)
            g_launch_timer = LaunchTimer{events_[4 * timed + 2], events_[4 * timed + 3]};
        if (int rc = launch_backproject_update(x_[cur_].as<float>(), x_[cur_ ^ 1].as<float>(), (int)R, (int)C, ch,
                                               fwd, fastB ? &dfwd : nullptr, sg, K, s, gauss_, btv_, p.tau, p.lambda,
                                               st, still))
            return rc;
        if (time_it)
        {
            g_launch_timer = LaunchTimer{};
            ++timed;
        }
        cur_ ^= 1;
    }
    if (timing_)
        pending_iters_ = timed;

    if (out)
        if (int rc = launch_crop_convert(x_[cur_].as<float>(), (int)R, (int)C, ch, B, out, SRX_32F, ps_, st))
            return rc;
    if (x_full)
    {
        if (ps_)
        {
            if (int rc = launch_relayout(x_[cur_].as<float>(), x_full, (size_t)R * C, ch, ps_, false, st))
                return rc;
        }
        else
            SRX_HIP(hipMemcpyAsync(x_full, x_[cur_].get(), (size_t)R * C * ch * sizeof(float),
                                   hipMemcpyDeviceToDevice, st));
    }
    return SRX_OK;
}

int Solver::process(const float* const* y, const float* const* fwd_motions, int K, int base_idx, int rows, int cols,
                    int ch, const Params& p, float* out, float* x_full, hipStream_t st)
{
    // announced byte frames are for this call only, also when it fails before it gets to run()
    struct Forget
    {
        int& count;
        bool armed = true;
        ~Forget()
        {
            if (armed)
                count = 0;
        }
    } forget{y8_count_};
    if (K < 1 || K > kMaxWindow || base_idx < 0 || base_idx >= K)
    {
        set_error("process: window size or base index out of range");
        return SRX_ERR_INVALID;
    }
    if (int rc = validate_params(p))
        return rc;
    if (int rc = ensure(K, rows, cols, ch, p.scale, st))
        return rc;
    PtrList f{};
    MutPtrList r{};
    const float* relp[kMaxWindow];
    for (int k = 0; k < K; ++k)
    {
        if (k < K - 1)
            f.p[k] = fwd_motions[k];
        r.p[k] = rel_[k].get();
        relp[k] = rel_[k].as<float>();
    }
    if (int rc = launch_relative_motions(f, r, K, base_idx, (size_t)rows * cols * 2, st))
        return rc;
    forget.armed = false;
    still_frame_ = base_idx; // k_relative_motions has just written zeros as the base frame's motion
    return run(y, relp, K, base_idx, rows, cols, ch, p, out, x_full, st);
}

} // namespace srx
