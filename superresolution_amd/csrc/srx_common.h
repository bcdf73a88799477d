// Internal declarations shared by the host driver and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <string>

namespace srx
{

constexpr int kMaxWindow = 40;      // frames per temporal window (2*radius+1 <= 40)
constexpr int kMaxGauss = 31;       // largest blur kernel size
constexpr int kMaxBtvRadius = 7;    // btvKernelSize <= 15
constexpr int kMaxBtvWeights = 96;  // sum_{m=0..7} (7+m+1) = 92

// Kernel argument blocks passed by value.
struct PtrList
{
    const void* p[kMaxWindow];
};
struct MutPtrList
{
    void* p[kMaxWindow];
};
struct GaussK
{
    float k[kMaxGauss + 1];
    int n;
};
struct BtvW
{
    float w[kMaxBtvWeights];
    int radius; // (btvKernelSize - 1) / 2
};

// Host-built tap tables of cv::resize(INTER_CUBIC): per destination column /
// row the source index of tap 1 and the four weights.
struct CubicTables
{
    const int* xofs;
    const float4* alpha;
    const int* yofs;
    const float4* beta;
};

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);

#define SRX_HIP(call)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e__ = (call);                                                                                       \
        if (e__ != hipSuccess)                                                                                         \
            return ::srx::hip_fail(e__, #call);                                                                        \
    } while (0)

// ---- launchers (srx_kernels.hip) -----------------------------------------
int launch_relative_motions(const PtrList& fwd, const MutPtrList& rel, int n, int base_idx, size_t count,
                            hipStream_t st);
// dec_fwd / dec_bwd (optional, scale 4 only): decoded maps for the specialised kernels;
// dec_bwd holds dec_bwd_words(hr pixels) words: the group map in its short and its long form (one and two words per four pixels) and
// behind it one flag byte per LR row and strip of the forward kernel (fewer than hr_px / 16)
inline size_t dec_bwd_words(size_t hr_px)
{
    return 3 * (hr_px / 4) + hr_px / 64 + 4;
}
// dec_fwd holds dec_fwd_words(...) words: one entry per HR pixel and, at scale 2, behind them one word per 64 x 16
// tile of the update kernel (where the tile's sign cells lie; an even number of words) and the 16-bit form of
// the entries, relative to the tiles (see k_tile_relative)
constexpr int kUpdTileCols = 64, kUpdTileRows = 16;
inline size_t dec_fwd_tile_words(int hr_rows, int hr_cols)
{
    const size_t tiles = (size_t)((hr_cols + kUpdTileCols - 1) / kUpdTileCols) * ((hr_rows + kUpdTileRows - 1) / kUpdTileRows);
    return (tiles + 1) & ~(size_t)1;
}
inline size_t dec_fwd_words(int hr_rows, int hr_cols, int scale)
{
    const size_t px = (size_t)hr_rows * hr_cols;
    return px + (scale == 2 ? dec_fwd_tile_words(hr_rows, hr_cols) + (px + 1) / 2 : 0);
}
int launch_build_maps(const float* rel, int rows, int cols, int scale, const CubicTables& t, int16_t* fwd_map,
                      int16_t* bwd_map, uint32_t* dec_fwd, uint32_t* dec_bwd, hipStream_t st);
int launch_decode_maps(const int16_t* fwd_map, const int16_t* bwd_map, int hr_rows, int hr_cols, int scale,
                       uint32_t* dec_fwd, uint32_t* dec_bwd, hipStream_t st);
// plane_stride == 0: interleaved HR image; otherwise planar (plane c at c * plane_stride floats)
int launch_resize_cubic(const float* src, int srows, int scols, int ch, const CubicTables& t, float* dst, int drows,
                        int dcols, size_t plane_stride, hipStream_t st);
int launch_relayout(const float* src, float* dst, size_t pixels, int ch, size_t plane_stride, bool to_planar,
                    hipStream_t st);
inline size_t plane_stride_for(size_t pixels)
{
    return pixels + 4; // keeps planes 16-byte aligned and leaves room for the zero pixel at index `pixels`
}
int launch_btv_regularization(const float* src, int rows, int cols, int ch, const BtvW& w, float* dst, hipStream_t st);
// Which parameter sets the specialised kernels cover.
bool forward_fast_eligible(int hr_rows, int hr_cols, int ch, int scale, const GaussK& g);
bool backproject_fast_eligible(int hr_rows, int hr_cols, int ch, int scale, const GaussK& g, const BtvW& w);
// Sign planes: the generic kernels use one word per LR pixel, the specialised ones a pair
// plane of two words per LR pixel, {S(y, x), S(y+1, x)}, plus one spare cell.
inline size_t sign_pair_bytes(int lr_rows, int lr_cols)
{
    return ((size_t)lr_rows * lr_cols + 1) * 8;
}
int launch_sign_pairs(const uint32_t* plain, uint32_t* pairs, int lr_rows, int lr_cols, hipStream_t st);
// 0: specialised / tiled kernels wherever the parameters allow (default); 1: the plain per-pixel kernels only
extern int g_kernel_path;

int launch_sign_unpair(const uint32_t* pairs, uint32_t* plain, int lr_rows, int lr_cols, hipStream_t st);
// sign(y_k - D H M_k X) for K frames in one launch.  With dec_bwd (and an
// eligible parameter set) the specialised kernel runs; it wants X planar
// (plane_stride_for) with the zero pixel at index hr_rows*hr_cols of every plane
// and writes pair planes.  Otherwise the generic kernel runs: plain sign planes, X interleaved
// (generic_plane_stride 0) or planar with that plane stride.
// y_bytes (optional): the same LR frames as 8-bit originals (rows x cols x ch bytes each); the specialised
// scale-4 and scale-2 kernels then read those instead of the float copies (same values, a quarter of the bytes).
// still_frame (optional hint, -1 = none): a frame whose motion is zero (the base frame of a window).  It only
// decides which wave of the specialised scale-4 kernel works on which frame; results do not depend on it.
int launch_forward_sign(const float* X, int hr_rows, int hr_cols, int ch, const PtrList& raw_bwd,
                        const PtrList* dec_bwd, const PtrList& y, const MutPtrList& signs, int K, int scale,
                        const GaussK& g, size_t generic_plane_stride, hipStream_t st,
                        const PtrList* y_bytes = nullptr, int still_frame = -1, const void* tile_plan = nullptr);
// Events for the next launch of launch_forward_sign / launch_backproject_update on this thread (both null: none):
// start is tied to the kernel's begin, stop to its end.  Cleared by the caller after the launch.
struct LaunchTimer
{
    hipEvent_t start = nullptr, stop = nullptr;
};
extern thread_local LaunchTimer g_launch_timer;

// The specialised scale-4 forward kernel's tiles cut by the cost of their rows (rows whose gathers leave the
// kernel's LDS window are slow): launch_plan_forward_tiles reads the row kinds that launch_build_maps left behind
// the K frames' decoded backward maps and writes the plan (forward_plan_bytes; scratch of *scratch_bytes) that
// launch_forward_sign then takes as tile_plan.  *planned = false: no plan was made (tiny images), pass nullptr.
size_t forward_plan_bytes(int hr_rows, int hr_cols, int ch, int K, size_t* scratch_bytes);
int launch_plan_forward_tiles(const PtrList& dec_bwd, int K, int hr_rows, int hr_cols, int ch, void* plan, void* scratch,
                              bool* planned, hipStream_t st);
// With dec_fwd (and an eligible parameter set) the specialised kernel runs; it
// wants X and X_out planar and the signs as pair planes (sign_pair_bytes).
// identity_frame (-1 = none): a frame whose forward map is known to be the identity (zero motion: the base frame
// of a window built by calcRelativeMotions).  The specialised scale-4 kernel then derives that frame's entries
// from the pixel's own place instead of reading them; the caller vouches for the map, the values are the same.
int launch_backproject_update(const float* X, float* X_out, int hr_rows, int hr_cols, int ch, const PtrList& raw_fwd,
                              const PtrList* dec_fwd, const PtrList& signs, int K, int scale, const GaussK& g,
                              const BtvW& w, double tau, double lambda, hipStream_t st, int identity_frame = -1);
int launch_crop_convert(const float* X, int rows, int cols, int ch, int border, void* dst, int depth,
                        size_t plane_stride, hipStream_t st);
int launch_convert_8u_to_32f(const uint8_t* src, size_t count, float* dst, hipStream_t st);
int launch_convert_16u_to_32f(const uint16_t* src, size_t count, float* dst, hipStream_t st);

} // namespace srx
