// Host-side classes of the srx library: device buffers, the BTV-L1 solver and
// the streaming driver behind the C ABI of include/srx.h.
#pragma once

#include "srx_common.h"
#include "../../include/srx.h"

#include <array>
#include <memory>
#include <vector>

namespace srx
{

// Grow-only device allocation bound to the device that was current at the
// first reserve().
class DeviceBuffer
{
  public:
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    DeviceBuffer(DeviceBuffer&& o) noexcept : ptr_(o.ptr_), cap_(o.cap_) { o.ptr_ = nullptr, o.cap_ = 0; }
    DeviceBuffer& operator=(DeviceBuffer&& o) noexcept
    {
        if (this != &o)
        {
            release();
            ptr_ = o.ptr_, cap_ = o.cap_;
            o.ptr_ = nullptr, o.cap_ = 0;
        }
        return *this;
    }
    ~DeviceBuffer() { release(); }
    int reserve(size_t bytes);
    void release();
    template <typename T> T* as() const { return static_cast<T*>(ptr_); }
    void* get() const { return ptr_; }
    size_t capacity() const { return cap_; }

  private:
    void* ptr_ = nullptr;
    size_t cap_ = 0;
};

class PinnedBuffer
{
  public:
    PinnedBuffer() = default;
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    ~PinnedBuffer() { release(); }
    int reserve(size_t bytes);
    void release();
    void* get() const { return ptr_; }

  private:
    void* ptr_ = nullptr;
    size_t cap_ = 0;
};

// Pitched copy that collapses to one linear copy when both sides are dense: a 2-D copy from
// pageable host memory is carried out row by row by the runtime (milliseconds for a 540-row frame).
inline hipError_t copy_pitched(void* dst, size_t dpitch, const void* src, size_t spitch, size_t rowbytes, size_t rows,
                               hipMemcpyKind kind, hipStream_t st)
{
    if (dpitch == rowbytes && spitch == rowbytes)
        return hipMemcpyAsync(dst, src, rowbytes * rows, kind, st);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, rowbytes, rows, kind, st);
}

// Parameters of BTV_L1_Base / BTV_L1 (lib/include/super_resolution.hpp:54-62,172)
struct Params
{
    int scale = 4;
    int iterations = 180;
    double tau = 1.3;
    double lambda = 0.03;
    double alpha = 0.7;
    int btvKernelSize = 7;
    int blurKernelSize = 5;
    double blurSigma = 0.0;
    int temporalAreaRadius = 4;
};

const std::string& last_error();


// Validates what the reference only debug-asserts (lib/src/btv_l1.cpp:359-365).
int validate_params(const Params& p);

// Quality metrics (srx_quality.hip): cv::PSNR and the MSSIM of samples/quality_test/main.cpp:140-199.
// Images may live in host or device memory; the calls return when the result is on the host.
int quality_psnr(const srx_image& a, const srx_image& b, double* psnr, hipStream_t st);
int quality_psnr_noalloc(const srx_image& a, const srx_image& b, double* psnr, hipStream_t st);
int quality_mssim(const srx_image& a, const srx_image& b, double* per_channel, double* mssim, hipStream_t st);

// Host-built cv::resize(INTER_CUBIC) tap tables
void cubic_table_x(int scols, int dcols, std::vector<int>& sx, std::vector<float>& alpha);
void cubic_table_y(int srows, int drows, std::vector<int>& sy, std::vector<float>& beta);
int gaussian_kernel(int n, double sigma, float* k);
int calc_btv_weights(int btv_kernel_size, double alpha, float* w);

// Device copy of the tap tables for one (src, dst) geometry.
class CubicTablesDev
{
  public:
    int ensure(int srows, int scols, int drows, int dcols, hipStream_t st);
    CubicTables view() const;

  private:
    int srows_ = 0, scols_ = 0, drows_ = 0, dcols_ = 0;
    DeviceBuffer xofs_, alpha_, yofs_, beta_;
};

// BTV_L1_Base::run on device-resident inputs (lib/src/btv_l1.cpp:357-457).
class Solver
{
  public:
    Solver() = default;
    Solver(const Solver&) = delete;
    Solver& operator=(const Solver&) = delete;
    ~Solver() { release(); }
    int run(const float* const* y, const float* const* rel, int K, int base_idx, int rows, int cols, int ch,
            const Params& p, float* out, float* x_full, hipStream_t st);
    int process(const float* const* y, const float* const* fwd_motions, int K, int base_idx, int rows, int cols,
                int ch, const Params& p, float* out, float* x_full, hipStream_t st);
    // The 8-bit originals of the K frames of the next run()/process() call (device, rows x cols x ch bytes each,
    // dense): when the frames came in as 8-bit images the float copies passed to run() hold the same values, and
    // the specialised forward kernel reads the bytes instead (a quarter of the traffic).  Valid for one call.
    void byte_frames(const uint8_t* const* y8, int K)
    {
        y8_count_ = 0;
        if (y8 && K >= 1 && K <= kMaxWindow)
        {
            for (int k = 0; k < K; ++k)
                y8_[k] = y8[k];
            y8_count_ = K;
        }
    }
    // HR image after the last run (device), rows*scale x cols*scale x ch; planar with this plane
    // stride when last_x_plane_stride() != 0, interleaved otherwise
    const float* last_x() const { return x_[cur_].as<float>(); }
    size_t last_x_plane_stride() const { return ps_; }
    void release();

    // Optional per-kernel timing with HIP events on the solver's stream: the forward
    // and back-projection launches of every kTimingStride-th iteration carry a start and
    // a stop event each (the kernel's own begin and end), read back (after a stream
    // synchronise) by collect_timing().
    static constexpr int kTimingStride = 8;
    void enable_timing(bool on) { timing_ = on; }
    bool timing() const { return timing_; }
    int collect_timing(hipStream_t st);
    void reset_timing() { fwd_ms_ = bwd_ms_ = 0.0, timed_iters_ = 0; }
    double fwd_ms() const { return fwd_ms_; }
    double bwd_ms() const { return bwd_ms_; }
    long timed_iterations() const { return timed_iters_; }

  private:
    int ensure(int K, int rows, int cols, int ch, int scale, hipStream_t st);
    std::vector<DeviceBuffer> fwd_maps_, bwd_maps_, dec_fwd_, dec_bwd_, signs_, rel_;
    DeviceBuffer x_[2];
    DeviceBuffer tile_plan_, tile_cost_; // the forward kernel's tiles for the current window's maps
    const uint8_t* y8_[kMaxWindow] = {};
    int y8_count_ = 0;
    int still_frame_ = -1; // window frame whose relative motion process() has just set to zero (else -1)
    int cur_ = 0;
    size_t ps_ = 0; // plane stride of x_ (0: interleaved)
    CubicTablesDev tables_;
    // cached filter / weights, re-derived on parameter change (lib/src/btv_l1.cpp:392-405)
    GaussK gauss_{};
    int curBlurKernelSize_ = -1;
    double curBlurSigma_ = -1.0;
    BtvW btv_{};
    int curBtvKernelSize_ = -1;
    double curAlpha_ = -1.0;
    bool timing_ = false;
    std::vector<hipEvent_t> events_; // 4 per timed iteration of the last run
    int pending_iters_ = 0;
    double fwd_ms_ = 0.0, bwd_ms_ = 0.0;
    long timed_iters_ = 0;
};

} // namespace srx
