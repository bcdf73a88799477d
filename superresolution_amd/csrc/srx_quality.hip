// Quality metrics of the reference's quality harness on the GPU: cv::PSNR and
// the MSSIM of samples/quality_test/main.cpp:140-199.
//
// Both are reductions over the image: every thread adds up its share, a wave
// adds its 64 lanes with cross-lane shuffles, one partial per block goes to
// memory, and the (few thousand) partials are added on the host in index order,
// so a result does not depend on how blocks were scheduled.
//
// Build with -ffp-contract=off: the SSIM map follows the float operations of
// the reference one by one.
#include "srx_host.h"

#include <cfloat>
#include <cmath>
#include <mutex>
#include <vector>

namespace srx
{

namespace
{

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while ((unsigned)p >= (unsigned)len)
        p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off);
    return v; // lane 0 holds the sum
}

// Sum of squared differences of two 8-bit images with row pitches; one row
// segment of 4096 bytes per block, 16 bytes per thread, 256 apart.
constexpr int kSegBytes = 4096;
__global__ __launch_bounds__(256) void k_sqdiff_8u(const uint8_t* __restrict__ a, size_t a_step,
                                                    const uint8_t* __restrict__ b, size_t b_step, int rowbytes,
                                                    int segs_per_row, unsigned long long* __restrict__ partial)
{
    __shared__ unsigned wsum[4];
    const int row = blockIdx.x / segs_per_row, seg = blockIdx.x - row * segs_per_row;
    const uint8_t* pa = a + (size_t)row * a_step;
    const uint8_t* pb = b + (size_t)row * b_step;
    unsigned s = 0;
#pragma unroll 4
    for (int i = 0; i < kSegBytes / 256; ++i)
    {
        const int x = seg * kSegBytes + i * 256 + threadIdx.x;
        if (x < rowbytes)
        {
            const int d = (int)pa[x] - (int)pb[x];
            s += (unsigned)(d * d);
        }
    }
    s = wave_sum(s); // <= 4096 * 65025 per block: fits 32 bits
    if ((threadIdx.x & 63) == 0)
        wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[blockIdx.x] = (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Partial sums of the variant of quality_psnr that allocates nothing: kNoAllocMaxBlocks partials, then one
// 256-byte slot for the total.  Every GPU the library runs on holds its own copy of this array (520 KiB) at
// its own address until the process ends; host code reaches it only through the per-device address taken with
// hipGetSymbolAddress (the array's name is not a device address on the host side).  The array keeps the
// contents of the last call; a call overwrites every element it reads.
constexpr size_t kNoAllocMaxBlocks = 65536;
constexpr size_t kNoAllocTotalSlot = kNoAllocMaxBlocks;
static __device__ __align__(256) unsigned long long scratch[kNoAllocMaxBlocks + 32];

constexpr int kMaxScratchDevices = 64;
std::mutex g_scratch_mutex;
unsigned long long* g_scratch_ptr[kMaxScratchDevices] = {};

// The GPU that owns a device pointer, or an error for anything that is not plain device memory.
int device_of_pointer(const void* p, const char* which, int* dev)
{
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, p) != hipSuccess || attr.type != hipMemoryTypeDevice)
    {
        (void)hipGetLastError();
        set_error(std::string("psnr (no allocation): image ") + which + " is not in device memory");
        return SRX_ERR_INVALID;
    }
    *dev = attr.device;
    return SRX_OK;
}

// Address of `scratch` on the current device, looked up once per device.  Both images must live on that device:
// every GPU has its own copy of the array at its own address, and the caller's stream belongs to the current
// device, so an image of another GPU is refused rather than handed a foreign array.
int scratch_of_current_device(const srx_image& a, const srx_image& b, unsigned long long** out)
{
    int dev = -1, dev_a = -1, dev_b = -1;
    SRX_HIP(hipGetDevice(&dev));
    if (int rc = device_of_pointer(a.data, "a", &dev_a))
        return rc;
    if (int rc = device_of_pointer(b.data, "b", &dev_b))
        return rc;
    if (dev_a != dev || dev_b != dev)
    {
        set_error("psnr (no allocation): the images live on GPU " + std::to_string(dev_a) + " and GPU " +
                  std::to_string(dev_b) + " but the current device is GPU " + std::to_string(dev) +
                  " (make the images' device current and pass one of its streams)");
        return SRX_ERR_INVALID;
    }
    if (dev < 0 || dev >= kMaxScratchDevices)
    {
        set_error("psnr (no allocation): device index out of range");
        return SRX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev])
    {
        void* p = nullptr;
        SRX_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p)
        {
            set_error("psnr (no allocation): the scratch array has no address on this device");
            return SRX_ERR_DEVICE;
        }
        g_scratch_ptr[dev] = static_cast<unsigned long long*>(p);
    }
    *out = g_scratch_ptr[dev];
    return SRX_OK;
}

// Adds `n` partials into *total (one block; integer sums, so the order is free).
__global__ __launch_bounds__(256) void k_sum_partials(const unsigned long long* __restrict__ partial, unsigned n,
                                                       unsigned long long* __restrict__ total)
{
    __shared__ unsigned long long wsum[4];
    unsigned long long s = 0;
    for (unsigned i = threadIdx.x; i < n; i += 256)
        s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0)
        wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

struct Gauss11
{
    float k[11];
};

template <typename T> __device__ __forceinline__ float as_float(T v)
{
    return (float)v; // convertTo(CV_32F): exact for 8-bit
}

// SSIM map of one 32x16 tile, summed per channel.
//  * the tile of both images (+5 px, reflect101) is staged in LDS as floats;
//  * row pass of the 11-tap Gaussian over I1, I2, I1*I1, I2*I2, I1*I2 (the three
//    products are rounded floats, as the reference's Mat::mul results are), taps
//    accumulated left to right as cv::RowFilter does for kernels wider than 5;
//  * column pass in cv::SymmColumnFilter order: centre, then symmetric pairs;
//  * the SSIM expression in the reference's operation order, summed in double.
constexpr int kTX = 32, kTY = 16, kR = 5;
constexpr int kLW = kTX + 2 * kR, kLH = kTY + 2 * kR;
template <typename T>
__global__ __launch_bounds__(256) void k_ssim_sum(const T* __restrict__ a, size_t a_step, const T* __restrict__ b,
                                                   size_t b_step, int rows, int cols, int ch, Gauss11 g, int tiles_x,
                                                   double* __restrict__ partial)
{
    __shared__ float in1[kLH * kLW], in2[kLH * kLW];
    __shared__ float rowp[5][kLH * kTX];
    __shared__ double wsum[4][4];
    const int tile_x = blockIdx.x % tiles_x, tile_y = blockIdx.x / tiles_x;
    const int x0 = tile_x * kTX, y0 = tile_y * kTY;
    const int tid = threadIdx.x;
    const float C1 = (float)6.5025, C2 = (float)58.5225;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};

    for (int c = 0; c < ch; ++c)
    {
        for (int idx = tid; idx < kLH * kLW; idx += 256)
        {
            const int r = idx / kLW, cc = idx - r * kLW;
            const int gy = reflect101(y0 - kR + r, rows), gx = reflect101(x0 - kR + cc, cols);
            in1[idx] = as_float(((const T*)((const char*)a + (size_t)gy * a_step))[gx * ch + c]);
            in2[idx] = as_float(((const T*)((const char*)b + (size_t)gy * b_step))[gx * ch + c]);
        }
        __syncthreads();
        for (int idx = tid; idx < kLH * kTX; idx += 256)
        {
            const int r = idx / kTX, x = idx - r * kTX;
            const float* p1 = in1 + r * kLW + x;
            const float* p2 = in2 + r * kLW + x;
            float s1 = g.k[0] * p1[0], s2 = g.k[0] * p2[0];
            float s11 = g.k[0] * (p1[0] * p1[0]), s22 = g.k[0] * (p2[0] * p2[0]), s12 = g.k[0] * (p1[0] * p2[0]);
#pragma unroll
            for (int j = 1; j < 11; ++j)
            {
                const float v1 = p1[j], v2 = p2[j];
                s1 += g.k[j] * v1;
                s2 += g.k[j] * v2;
                s11 += g.k[j] * (v1 * v1);
                s22 += g.k[j] * (v2 * v2);
                s12 += g.k[j] * (v1 * v2);
            }
            rowp[0][idx] = s1, rowp[1][idx] = s2, rowp[2][idx] = s11, rowp[3][idx] = s22, rowp[4][idx] = s12;
        }
        __syncthreads();
        for (int o = tid; o < kTY * kTX; o += 256)
        {
            const int y = o / kTX, x = o - y * kTX;
            float q[5];
#pragma unroll
            for (int i = 0; i < 5; ++i)
            {
                const float* R = rowp[i] + (y + kR) * kTX + x;
                float s = R[0] * g.k[kR];
#pragma unroll
                for (int j = 1; j <= kR; ++j)
                    s += (R[-j * kTX] + R[j * kTX]) * g.k[kR + j];
                q[i] = s;
            }
            const float mu1 = q[0], mu2 = q[1];
            const float mu1_2 = mu1 * mu1, mu2_2 = mu2 * mu2, mu1_mu2 = mu1 * mu2;
            const float sigma1_2 = q[2] - mu1_2, sigma2_2 = q[3] - mu2_2, sigma12 = q[4] - mu1_mu2;
            float t1 = mu1_mu2 * 2.0f + C1;
            float t2 = sigma12 * 2.0f + C2;
            const float numerator = t1 * t2;
            t1 = (mu1_2 + mu2_2) + C1;
            t2 = (sigma1_2 + sigma2_2) + C2;
            const float denominator = t1 * t2;
            const float ssim = denominator != 0.0f ? __fdiv_rn(numerator, denominator) : 0.0f; // cv::divide: 0 where the divisor is 0
            if (y0 + y < rows && x0 + x < cols)
                acc[c] += (double)ssim;
        }
        __syncthreads(); // the tiles are rewritten for the next channel
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
    {
        const double s = wave_sum(acc[c]);
        if ((tid & 63) == 0)
            wsum[tid >> 6][c] = s;
    }
    __syncthreads();
    if (tid < 4)
        partial[(size_t)blockIdx.x * 4 + tid] = ((wsum[0][tid] + wsum[1][tid]) + wsum[2][tid]) + wsum[3][tid];
}

int check_pair(const srx_image& a, const srx_image& b, const char* what)
{
    if (!a.data || !b.data || a.rows < 1 || a.cols < 1 || a.rows != b.rows || a.cols != b.cols ||
        a.channels != b.channels || a.depth != b.depth || a.channels < 1 || a.channels > 4 ||
        (a.depth != SRX_8U && a.depth != SRX_32F))
    {
        set_error(std::string(what) + ": the two images must have the same size, depth (8U or 32F) and 1..4 channels");
        return SRX_ERR_INVALID;
    }
    const size_t rowbytes = (size_t)a.cols * a.channels * (a.depth == SRX_8U ? 1 : 4);
    if ((a.step && a.step < rowbytes) || (b.step && b.step < rowbytes))
    {
        set_error(std::string(what) + ": step smaller than a row");
        return SRX_ERR_INVALID;
    }
    return SRX_OK;
}

// device view of an image: host images are uploaded (densely) into `tmp`
int device_view(const srx_image& im, DeviceBuffer& tmp, const void** ptr, size_t* step, hipStream_t st)
{
    const size_t rowbytes = (size_t)im.cols * im.channels * (im.depth == SRX_8U ? 1 : 4);
    if (im.mem == SRX_MEM_DEVICE)
    {
        *ptr = im.data;
        *step = im.step ? im.step : rowbytes;
        return SRX_OK;
    }
    if (int rc = tmp.reserve(rowbytes * im.rows))
        return rc;
    SRX_HIP(copy_pitched(tmp.get(), rowbytes, im.data, im.step ? im.step : rowbytes, rowbytes, im.rows,
                         hipMemcpyHostToDevice, st));
    *ptr = tmp.get();
    *step = rowbytes;
    return SRX_OK;
}

} // namespace

int quality_psnr(const srx_image& a, const srx_image& b, double* psnr, hipStream_t st)
{
    if (int rc = check_pair(a, b, "psnr"))
        return rc;
    if (a.depth != SRX_8U)
    {
        set_error("psnr: 8-bit images only (cv::PSNR asserts CV_8U)");
        return SRX_ERR_INVALID;
    }
    DeviceBuffer ta, tb, part;
    const void *pa, *pb;
    size_t sa, sb;
    if (int rc = device_view(a, ta, &pa, &sa, st))
        return rc;
    if (int rc = device_view(b, tb, &pb, &sb, st))
        return rc;
    const int rowbytes = a.cols * a.channels;
    const int segs = (rowbytes + kSegBytes - 1) / kSegBytes;
    const size_t blocks = (size_t)segs * a.rows;
    if (int rc = part.reserve(blocks * sizeof(unsigned long long)))
        return rc;
    hipLaunchKernelGGL(k_sqdiff_8u, dim3((unsigned)blocks), dim3(256), 0, st, (const uint8_t*)pa, sa, (const uint8_t*)pb,
                       sb, rowbytes, segs, part.as<unsigned long long>());
    SRX_HIP(hipGetLastError());
    std::vector<unsigned long long> h(blocks);
    SRX_HIP(hipMemcpyAsync(h.data(), part.get(), blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SRX_HIP(hipStreamSynchronize(st));
    unsigned long long s = 0;
    for (unsigned long long v : h)
        s += v;
    // cv::PSNR: sqrt(norm(a, b, NORM_L2SQR) / (total * channels)); 20 log10(255 / (diff + DBL_EPSILON))
    const double diff = std::sqrt((double)s / ((double)a.rows * a.cols * a.channels));
    *psnr = 20.0 * std::log10(255.0 / (diff + DBL_EPSILON));
    return SRX_OK;
}

// quality_psnr for callers that cannot allocate (code under stream capture set-up, tight serving loops, harnesses
// that forbid hipMalloc): device images only, partial sums in the per-device static array, the total formed on
// the device.  Runtime calls: two kernel launches, one 8-byte copy to the host, one stream synchronise.
// Calls that overlap on one GPU share the array and race.
int quality_psnr_noalloc(const srx_image& a, const srx_image& b, double* psnr, hipStream_t st)
{
    if (int rc = check_pair(a, b, "psnr (no allocation)"))
        return rc;
    if (a.depth != SRX_8U)
    {
        set_error("psnr (no allocation): 8-bit images only (cv::PSNR asserts CV_8U)");
        return SRX_ERR_INVALID;
    }
    if (a.mem != SRX_MEM_DEVICE || b.mem != SRX_MEM_DEVICE)
    {
        set_error("psnr (no allocation): both images must be in device memory (a host image needs a staging buffer)");
        return SRX_ERR_INVALID;
    }
    const int rowbytes = a.cols * a.channels;
    const int segs = (rowbytes + kSegBytes - 1) / kSegBytes;
    const size_t blocks = (size_t)segs * a.rows;
    if (blocks > kNoAllocMaxBlocks)
    {
        set_error("psnr (no allocation): " + std::to_string(blocks) + " row segments of 4096 bytes, the limit is " +
                  std::to_string(kNoAllocMaxBlocks) + " (use srx_quality_psnr)");
        return SRX_ERR_INVALID;
    }
    unsigned long long* part = nullptr;
    if (int rc = scratch_of_current_device(a, b, &part))
        return rc;
    const size_t rb = (size_t)rowbytes;
    hipLaunchKernelGGL(k_sqdiff_8u, dim3((unsigned)blocks), dim3(256), 0, st, (const uint8_t*)a.data,
                       a.step ? a.step : rb, (const uint8_t*)b.data, b.step ? b.step : rb, rowbytes, segs, part);
    SRX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, part, (unsigned)blocks, part + kNoAllocTotalSlot);
    SRX_HIP(hipGetLastError());
    unsigned long long s = 0;
    SRX_HIP(hipMemcpyAsync(&s, part + kNoAllocTotalSlot, sizeof(s), hipMemcpyDeviceToHost, st));
    SRX_HIP(hipStreamSynchronize(st));
    const double diff = std::sqrt((double)s / ((double)a.rows * a.cols * a.channels));
    *psnr = 20.0 * std::log10(255.0 / (diff + DBL_EPSILON));
    return SRX_OK;
}

int quality_mssim(const srx_image& a, const srx_image& b, double* per_channel, double* mssim, hipStream_t st)
{
    if (int rc = check_pair(a, b, "mssim"))
        return rc;
    DeviceBuffer ta, tb, part;
    const void *pa, *pb;
    size_t sa, sb;
    if (int rc = device_view(a, ta, &pa, &sa, st))
        return rc;
    if (int rc = device_view(b, tb, &pb, &sb, st))
        return rc;
    Gauss11 g;
    if (int rc = gaussian_kernel(11, 1.5, g.k)) // GaussianBlur(..., Size(11, 11), 1.5)
        return rc;
    const int tiles_x = (a.cols + kTX - 1) / kTX, tiles_y = (a.rows + kTY - 1) / kTY;
    const size_t blocks = (size_t)tiles_x * tiles_y;
    if (int rc = part.reserve(blocks * 4 * sizeof(double)))
        return rc;
    if (a.depth == SRX_8U)
        hipLaunchKernelGGL(k_ssim_sum<uint8_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const uint8_t*)pa, sa,
                           (const uint8_t*)pb, sb, a.rows, a.cols, a.channels, g, tiles_x, part.as<double>());
    else
        hipLaunchKernelGGL(k_ssim_sum<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)pa, sa,
                           (const float*)pb, sb, a.rows, a.cols, a.channels, g, tiles_x, part.as<double>());
    SRX_HIP(hipGetLastError());
    std::vector<double> h(blocks * 4);
    SRX_HIP(hipMemcpyAsync(h.data(), part.get(), blocks * 4 * sizeof(double), hipMemcpyDeviceToHost, st));
    SRX_HIP(hipStreamSynchronize(st));
    double m[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < blocks; ++i)
        for (int c = 0; c < 4; ++c)
            m[c] += h[4 * i + c];
    for (int c = 0; c < 4; ++c)
        m[c] = c < a.channels ? m[c] / ((double)a.rows * a.cols) : 0.0; // cv::mean
    if (per_channel)
        for (int c = 0; c < 4; ++c)
            per_channel[c] = m[c];
    // samples/quality_test/main.cpp:195-198, as written there: component 3 is read where 2 is meant
    if (mssim)
        *mssim = a.channels == 1 ? m[0] : (m[0] + m[1] + m[3]) / 3;
    return SRX_OK;
}

} // namespace srx
