// HIP kernels of the BTV-L1 solver for gfx950 (MI355X).
//
// Build with -ffp-contract=off: the solver's sign() steps make the iteration
// sensitive to the last bit, so every float operation below rounds exactly as
// written, in the order of the OpenCV routines the reference calls.
//
// Data layout in HBM
//   images        float32, dense rows.  At the C ABI and in the generic kernels
//                 channels are interleaved.  The specialised kernels keep the
//                 HR image planar: plane c starts at c * plane_stride floats,
//                 plane_stride = rows*cols + 4, and element rows*cols of every
//                 plane is an all-zero pixel (the value of cv::remap's constant
//                 border).  Planar gathers of neighbouring lanes hit
//                 consecutive addresses; the BTV tile needs no de-interleave.
//   raw maps      int16 (x, y) per HR pixel: the source index that
//                 cv::remap(INTER_NEAREST) derives from its float map.
//   decoded maps  one uint32 per HR pixel, derived from the raw maps for the
//                 specialised kernels:
//                   backward: the group map, two words per 4 pixels of a
//                             row (kGrp* below), laid out per LR pixel (y, x):
//                             first all "word 0", four per LR pixel, for HR
//                             rows 4y-1 .. 4y+2 (reflected at the top) and
//                             columns 4x .. 4x+3, then all "word 1" in the
//                             same order - the forward kernel reads an LR
//                             pixel's four rows with two 16-byte loads;
//                   forward:  see kFwd* below (scale 4) and decode_fwd2
//                             (scale 2).
//   signs         one uint32 word per LR pixel and window frame: sign+1 (0,1,2)
//                 of channel c in bits 10c..10c+1.  Generic and scale-2 kernels
//                 keep a plain plane of words (the scale-2 update kernel reads
//                 one spare row behind it).  The scale-4 kernels keep a "pair
//                 plane": cell (y, x) is the two words {S(y, x), S(y+1, x)},
//                 so that one 16-byte read at a cell returns the 2x2 block
//                 {S(y,x), S(y+1,x), S(y,x+1), S(y+1,x+1)}; rows*cols + 1
//                 cells.
//
// Several kernels compute the same values: per-pixel ones (any scale, blur
// size, BTV size: the in-library reference), a tiled forward kernel (any scale
// and blur size), and specialised pairs for scale 4 and scale 2 with the 5-tap
// blur and BTV size 7.
#include "srx_common.h"

#include <hip/hip_ext.h>


#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

// Diagnostic builds only (-DSRX_STAMPS): every wave of the update kernel records s_memtime at its phase
// boundaries into a device array that srx_debug_stamps copies out.  No stamp exists in the product build.
#ifdef SRX_STAMPS
constexpr int kStampSlots = 8;
__device__ unsigned long long g_stamps[kStampSlots << 16];
#define SRX_STAMP(slot)                                                                                       \
    do                                                                                                        \
    {                                                                                                         \
        if ((threadIdx.x & 63) == 0)                                                                          \
        {                                                                                                     \
            const unsigned wave_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                       \
            if (wave_ < (1u << 16))                                                                           \
                g_stamps[wave_ * kStampSlots + (slot)] = __builtin_amdgcn_s_memtime();                        \
        }                                                                                                     \
    } while (0)
extern "C" __attribute__((visibility("default"))) int srx_debug_stamps(unsigned long long* dst, size_t count)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), count * sizeof(unsigned long long));
}
#define SRX_TICK() __builtin_amdgcn_s_memtime()
#define SRX_FWD_STAMPS(v0, v1, v2, v3, v4)                                                                   \
    do                                                                                                        \
    {                                                                                                         \
        if ((threadIdx.x & 63) == 0)                                                                          \
        {                                                                                                     \
            const unsigned wave_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                       \
            if (wave_ < (1u << 16))                                                                           \
            {                                                                                                 \
                unsigned long long* o_ = g_stamps + wave_ * kStampSlots;                                      \
                o_[0] = (v0), o_[1] = (v1), o_[2] = (v2), o_[3] = (v3), o_[4] = (v4);                         \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
#else
#define SRX_STAMP(slot) do { } while (0)
#endif

namespace srx
{

namespace
{

// forward decoded-map entry (scale 4, 5-tap blur): a target (fx, fy) touches at
// most the 2x2 block of LR sign cells whose top-left cell is `cell`
//   bits 0..23   byte offset of the cell in the pair plane: 8 * (row*w + col)
//   bits 24..28  weight-table index px + 5*py, where px = fx & 3 (py likewise)
//                is the phase of the target on the LR grid; phase 4 stands for
//                fx = cols-2, whose tap at fx+2 falls on the reflected border
//                (reflect101 lands on a zero of the zero-inserted image, so the
//                tap drops out; the same holds for every tap reflected at the
//                left and top edges, which therefore need no entries of their
//                own); 25 = all weights 0 (target outside the image, or in the
//                last column / row, whose only tap is a reflected one)
// Group map of the backward map, two forms.  An entry describes the sources (col_i, row_i), i = 0..3, of 4
// neighbouring pixels of HR row N = 4y-1+r (the map row itself is the reflected one at the top edge), columns
// 4x..4x+3; four entries (r = 0..3) per LR pixel.
//  short form, one word (16 bytes per LR pixel, half of what the raw map takes for the same 16 pixels):
//   bits 0..6   col_0 - 4x, bits 7..13 row_0 - N, both signed (-63 .. 63);
//   for i = 1..3, at bit 14 + 6(i-1): col_i - (col_0 + i), then row_i - row_0 (3 bits each, signed: -4 .. 3);
//               all zero when the four sources are one run of pixels;
//   the value kGrpFar (a row field of -64, nothing else): some coordinate does not fit its field.
//  long form, two words in two planes:
//   word 0  bits 0..14 col_0 - 4x, bits 15..29 row_0 - N, both signed;
//           bit 30: "far" - some coordinate does not fit its field (read the raw map instead)
//   word 1  for i = 1..3, at bit 10(i-1): col_i - (col_0 + i) (5 bits, signed), then row_i - row_0
//           (5 bits, signed); all zero when the four sources are one run of pixels
// The forward kernel reads the short form for LR rows in which every lane's sources lie in its LDS window (nearly
// all rows: half the map traffic) and the long form for the others, where lanes go different ways.
// Sources outside the image keep their coordinates: the forward kernel stages zeros around the image.
constexpr uint32_t kGrpFar = 0x40u << 7;        // short form
constexpr uint32_t kGrpLongFar = 1u << 30;      // long form, word 0
constexpr int kGrpMaxDim = 16383;               // long form: displacements must fit 15 signed bits
template <bool LONG> struct Grp
{
    static __device__ __forceinline__ bool far(uint2 g) { return LONG ? (g.x & kGrpLongFar) != 0 : g.x == kGrpFar; }
    static __device__ __forceinline__ int dc(uint2 g) { return __builtin_amdgcn_sbfe((int)g.x, 0, LONG ? 15 : 7); }
    static __device__ __forceinline__ int dr(uint2 g) { return __builtin_amdgcn_sbfe((int)g.x, LONG ? 15 : 7, LONG ? 15 : 7); }
    static __device__ __forceinline__ int dx(uint2 g, int j)
    {
        return LONG ? __builtin_amdgcn_sbfe((int)g.y, 10 * (j - 1), 5) : __builtin_amdgcn_sbfe((int)g.x, 14 + 6 * (j - 1), 3);
    }
    static __device__ __forceinline__ int dy(uint2 g, int j)
    {
        return LONG ? __builtin_amdgcn_sbfe((int)g.y, 10 * (j - 1) + 5, 5) : __builtin_amdgcn_sbfe((int)g.x, 17 + 6 * (j - 1), 3);
    }
    // some source lies in another row than the first pixel's
    static __device__ __forceinline__ bool rows_differ(uint2 g)
    {
        return LONG ? (g.y & 0x3E0F83E0u) != 0 : (g.x & ((7u << 17) | (7u << 23) | (7u << 29))) != 0;
    }
    // the four sources are one run of pixels
    static __device__ __forceinline__ bool run(uint2 g) { return LONG ? g.y == 0 : (g.x & 0xFFFFC000u) == 0; }
};
constexpr uint32_t kFwdCellMask = 0x00FFFFFFu;
constexpr int kFwdLutShift = 24;
constexpr uint32_t kFwdLutZero = 25;
constexpr int kFwdLutSize = 26;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Loads and stores of data that is touched once per launch (maps, signs on their way out): marked
// non-temporal so that they do not push the lines of X, which every window frame gathers again, out of L2.
__device__ __forceinline__ uint4 stream_load(const uint4* p)
{
    const v4u v = __builtin_nontemporal_load((const v4u*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 stream_load(const uint2* p)
{
    const v2u v = __builtin_nontemporal_load((const v2u*)p);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void stream_store(uint32_t* p, uint32_t v)
{
    __builtin_nontemporal_store(v, p);
}
__device__ __forceinline__ void stream_store(uint2* p, uint32_t lo, uint32_t hi)
{
    __builtin_nontemporal_store(v2u{lo, hi}, (v2u*)p);
}

// volatile LDS vector access keeps 16-byte ds_read_b128 from being split into narrower reads
typedef __attribute__((address_space(3))) const volatile v4i lds_cv4i;

__device__ __forceinline__ int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len)
        return p;
    if (len == 1)
        return 0;
    do
    {
        p = p < 0 ? -p : 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi)
{
    return v < lo ? lo : (v > hi ? hi : v);
}

__device__ __forceinline__ float dsign(float a, float b)
{
    return a > b ? 1.0f : (a < b ? -1.0f : 0.0f);
}

// saturate_cast<short>(cvRound(v)): round half to even; NaN and values beyond
// the int range give SHRT_MIN (cvtsd2si's "integer indefinite", then saturated)
__device__ __forceinline__ short round_sat16(float v)
{
    if (!(fabsf(v) < 2147483648.0f))
        return (short)-32768;
    float r = rintf(v);
    r = fminf(fmaxf(r, -32768.0f), 32767.0f);
    return (short)(int)r;
}

// Blocks are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).
// Give each XCD one contiguous range of the work list so that neighbouring
// work items (which share halos and gather targets) meet in one L2.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned b, unsigned n)
{
    const unsigned xcd = b & 7u, local = b >> 3;
    const unsigned q = n >> 3, r = n & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// ---------------------------------------------------------------------------
// calcRelativeMotions (lib/src/btv_l1.cpp:36-57): running sums away from base
__global__ void k_relative_motions(PtrList fwd, MutPtrList rel, int n, int base_idx, size_t count)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count)
        return;
    ((float*)rel.p[base_idx])[e] = 0.0f;
    float acc = 0.0f;
    for (int i = base_idx - 1; i >= 0; --i)
    {
        acc = acc + ((const float*)fwd.p[i])[e];
        ((float*)rel.p[i])[e] = acc;
    }
    acc = 0.0f;
    for (int i = base_idx + 1; i < n; ++i)
    {
        acc = acc - ((const float*)fwd.p[i - 1])[e];
        ((float*)rel.p[i])[e] = acc;
    }
}

// ---------------------------------------------------------------------------
// One bicubic sample in cv::resize order: horizontal taps on 4 source rows
// (float intermediate), then the vertical taps.
template <int CH>
__device__ __forceinline__ void cubic_sample(const float* __restrict__ src, int srows, int scols, int sx, float4 a,
                                             int sy, float4 b, float (&out)[CH])
{
    const int x0 = clampi(sx - 1, 0, scols - 1), x1 = clampi(sx, 0, scols - 1);
    const int x2 = clampi(sx + 1, 0, scols - 1), x3 = clampi(sx + 2, 0, scols - 1);
    float h[4][CH];
#pragma unroll
    for (int r = 0; r < 4; ++r)
    {
        const float* S = src + (size_t)clampi(sy - 1 + r, 0, srows - 1) * scols * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            h[r][c] = S[x0 * CH + c] * a.x + S[x1 * CH + c] * a.y + S[x2 * CH + c] * a.z + S[x3 * CH + c] * a.w;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
        out[c] = h[0][c] * b.x + h[1][c] * b.y + h[2][c] * b.z + h[3][c] * b.w;
}

// decoded map entries (layout at the top of this file); w = C / 4 LR columns
__device__ __forceinline__ uint32_t decode_fwd(int fx, int fy, int R, int C)
{
    if ((unsigned)fx >= (unsigned)(C - 1) || (unsigned)fy >= (unsigned)(R - 1))
        return kFwdLutZero << kFwdLutShift;
    const int px = fx & 3, py = fy & 3;
    const int col = (fx >> 2) + (px == 3), row = (fy >> 2) + (py == 3);
    const int pxi = fx == C - 2 ? 4 : px, pyi = fy == R - 2 ? 4 : py;
    return (uint32_t)(row * (C >> 2) + col) * 8u | ((uint32_t)(pxi + 5 * pyi) << kFwdLutShift);
}

// Scale 2, 5-tap blur: a target touches at most 3x3 LR sign cells of a plain plane.
//   bits 0..23   byte offset of the top-left cell: 4 * (row*w + col)
//   bits 24..28  weight-table index px + 5*py; per axis: 0 even target (cells c-1, c, c+1 with k2, k0, k2),
//                1 odd target (cells c, c+1 with k1, k1), 2 target 0 (reflect101 folds the tap at -2 onto the
//                one at +2: k0, 2*k2), 3 last target (the tap beyond the edge folds back: 2*k1), 4 last but one
//                (k2, k0 + k2); 25 = all weights 0 (target outside the image)
__device__ __forceinline__ uint32_t decode_fwd2(int fx, int fy, int R, int C)
{
    if ((unsigned)fx >= (unsigned)C || (unsigned)fy >= (unsigned)R)
        return kFwdLutZero << kFwdLutShift;
    const int px = fx == 0 ? 2 : fx == C - 1 ? 3 : fx == C - 2 ? 4 : (fx & 1);
    const int py = fy == 0 ? 2 : fy == R - 1 ? 3 : fy == R - 2 ? 4 : (fy & 1);
    const int col = px == 0 ? (fx >> 1) - 1 : px == 2 ? 0 : px == 4 ? (fx >> 1) - 1 : (fx >> 1);
    const int row = py == 0 ? (fy >> 1) - 1 : py == 2 ? 0 : py == 4 ? (fy >> 1) - 1 : (fy >> 1);
    return (uint32_t)(row * (C >> 1) + col) * 4u | ((uint32_t)(px + 5 * py) << kFwdLutShift);
}

// upscaleMotions + buildMotionMaps + remap's float->short index conversion
// (lib/src/btv_l1.cpp:59-101; cv::remap INTER_NEAREST), plus the decoded maps
// for the specialised kernels when dec_fwd / dec_bwd are given (scale 4 only).
__global__ void k_build_maps(const float* __restrict__ rel, int rows, int cols, int scale, CubicTables t,
                             short2* __restrict__ fwd_map, short2* __restrict__ bwd_map, uint32_t* __restrict__ dec_fwd)
{
    const int R = rows * scale, C = cols * scale;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= C || y >= R)
        return;
    float m[2];
    cubic_sample<2>(rel, rows, cols, t.xofs[x], t.alpha[x], t.yofs[y], t.beta[y], m);
    const float s = (float)scale;
    const float mx = m[0] * s, my = m[1] * s;
    const int fx = round_sat16((float)x - mx), fy = round_sat16((float)y - my);
    const int bx = round_sat16((float)x + mx), by = round_sat16((float)y + my);
    const size_t o = (size_t)y * C + x;
    fwd_map[o] = make_short2((short)fx, (short)fy);
    bwd_map[o] = make_short2((short)bx, (short)by);
    if (dec_fwd)
        dec_fwd[o] = scale == 2 ? decode_fwd2(fx, fy, R, C) : decode_fwd(fx, fy, R, C);
}

// raw int16 maps -> decoded maps (same rules as k_build_maps), for callers that hold raw maps only
__global__ void k_decode_maps(const short2* __restrict__ fwd_map, int R, int C, int scale,
                              uint32_t* __restrict__ dec_fwd)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= C || y >= R)
        return;
    const size_t o = (size_t)y * C + x;
    dec_fwd[o] = scale == 2 ? decode_fwd2(fwd_map[o].x, fwd_map[o].y, R, C) : decode_fwd(fwd_map[o].x, fwd_map[o].y, R, C);
}

// Scale 2: the decoded forward map made relative to the update kernel's tiles.  A tile of 64 x 16 HR pixels
// back-projects from a small patch of every frame's sign plane (its own place shifted by the frame's motion),
// and the update kernel stages that patch in LDS instead of gathering 3 x 3 cells per pixel and frame from
// global memory (the texture path handles scattered dwords at an eighth of its coalesced rate, and those
// gathers were the kernel's bound).  One block per tile: finds the bounding box of the top-left cells of the
// tile's entries; if the boxes of all pixels fit a patch of kSgRows x kSgCols cells, the tile's entries are
// written a second time in a 16-bit form (behind the tile headers: half the map bytes for the update kernel to
// read), bits 0..9 = (row - row0) * kSgPitch + (col - col0), bits 10..14 = weight-table index, and the tile's
// header word (behind the map, one per tile) becomes 1 << 31 | byte offset of cell (row0, col0); otherwise the
// header is 0 (the update kernel then reads the 32-bit entries and gathers from global memory for this tile
// and frame).
constexpr int kSgRows = 12, kSgCols = 64, kSgPitch = 65, kSgWords = kSgRows * kSgPitch;
constexpr uint32_t kSgFit = 1u << 31, kSgCellMask = 1023u;
constexpr int kSgLutShift = 10;
__global__ __launch_bounds__(256) void k_tile_relative(uint32_t* __restrict__ dec, int H, int W, int tiles_x,
                                                       unsigned tile_words)
{
    __shared__ int lo_r, hi_r, lo_c, hi_c;
    const int t = blockIdx.x, tid = threadIdx.x;
    const int j0 = (t % tiles_x) * kUpdTileCols, i0 = (t / tiles_x) * kUpdTileRows;
    const int ty = tid >> 4, tx = tid & 15;
    const bool live = i0 + ty < H && j0 + 4 * tx < W; // W is a multiple of 4
    const int i = min(i0 + ty, H - 1), j = min(j0 + 4 * tx, W - 4);
    const int w = W >> 1;
    if (tid == 0)
        lo_r = lo_c = 1 << 30, hi_r = hi_c = -1;
    const uint4 e4 = *(const uint4*)(dec + (size_t)i * W + j);
    const uint32_t e[4] = {e4.x, e4.y, e4.z, e4.w};
    int row[4], col[4];
    bool used[4];
    __syncthreads();
    int mn_r = 1 << 30, mx_r = -1, mn_c = 1 << 30, mx_c = -1;
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
        used[q] = (e[q] >> kFwdLutShift) != kFwdLutZero;
        const int cell = (int)((e[q] & kFwdCellMask) >> 2);
        row[q] = cell / w, col[q] = cell - row[q] * w;
        if (used[q])
        {
            mn_r = min(mn_r, row[q]), mx_r = max(mx_r, row[q]);
            mn_c = min(mn_c, col[q]), mx_c = max(mx_c, col[q]);
        }
    }
    // one set of atomics per wave
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        mn_r = min(mn_r, __shfl_xor(mn_r, off)), mx_r = max(mx_r, __shfl_xor(mx_r, off));
        mn_c = min(mn_c, __shfl_xor(mn_c, off)), mx_c = max(mx_c, __shfl_xor(mx_c, off));
    }
    if ((tid & 63) == 0 && mx_r >= 0)
    {
        atomicMin(&lo_r, mn_r), atomicMax(&hi_r, mx_r);
        atomicMin(&lo_c, mn_c), atomicMax(&hi_c, mx_c);
    }
    __syncthreads();
    const bool any = hi_r >= 0;
    const int r0 = any ? lo_r : 0, c0 = any ? lo_c : 0;
    const bool fit = !any || (hi_r - lo_r + 3 <= kSgRows && hi_c - lo_c + 3 <= kSgCols);
    if (fit && live)
    {
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            o[q] = ((e[q] >> kFwdLutShift) & 31u) << kSgLutShift |
                   (used[q] ? (uint32_t)((row[q] - r0) * kSgPitch + (col[q] - c0)) : 0u);
        uint2* at16 = (uint2*)(dec + (size_t)H * W + tile_words) + ((size_t)i * W + j) / 4;
        *at16 = make_uint2(o[0] | o[1] << 16, o[2] | o[3] << 16);
    }
    if (tid == 0)
        dec[(size_t)H * W + t] = fit ? (kSgFit | (uint32_t)(r0 * w + c0) * 4u) : 0u;
}

// group map of the backward map (layout at the top of this file); one thread per entry
__global__ void k_group_maps(const short2* __restrict__ bwd_map, uint32_t* __restrict__ grp, int H, int W)
{
    const int w = W >> 2, h = H >> 2;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t entries = (size_t)h * w * 4;
    if (e >= entries)
        return;
    const int r = (int)(e & 3);
    const size_t cell = e >> 2;
    const int x = (int)(cell % w), y = (int)(cell / w);
    const int N = 4 * y - 1 + r;
    const short2* m = bwd_map + (size_t)reflect101(N, H) * W + 4 * x;
    const int col0 = m[0].x, row0 = m[0].y;
    {
        const int ddx = clampi(col0 - 4 * x, -63, 63), ddy = clampi(row0 - N, -63, 63);
        bool far = ddx != col0 - 4 * x || ddy != row0 - N;
        uint32_t others = 0;
        for (int i = 1; i <= 3; ++i)
        {
            const int dx = m[i].x - (col0 + i), dy = m[i].y - row0;
            far = far || dx < -4 || dx > 3 || dy < -4 || dy > 3;
            others |= ((uint32_t)(dx & 7) | ((uint32_t)(dy & 7) << 3)) << (6 * (i - 1));
        }
        grp[e] = far ? kGrpFar : (((uint32_t)ddx & 0x7Fu) | (((uint32_t)ddy & 0x7Fu) << 7) | (others << 14));
    }
    {
        const int ddx = clampi(col0 - 4 * x, -16384, 16383), ddy = clampi(row0 - N, -16384, 16383);
        bool far = ddx != col0 - 4 * x || ddy != row0 - N;
        uint32_t w1 = 0;
        for (int i = 1; i <= 3; ++i)
        {
            const int dx = m[i].x - (col0 + i), dy = m[i].y - row0;
            far = far || dx < -16 || dx > 15 || dy < -16 || dy > 15;
            w1 |= ((uint32_t)(dx & 31) | ((uint32_t)(dy & 31) << 5)) << (10 * (i - 1));
        }
        grp[entries + e] = ((uint32_t)ddx & 0x7FFFu) | (((uint32_t)ddy & 0x7FFFu) << 15) | (far ? kGrpLongFar : 0u);
        grp[2 * entries + e] = w1;
    }
}

// Geometry of the forward kernel's LDS window (described above k_forward_sign_lds)
constexpr int kFwdStrip = 63;
// 47 rows of three planes fill the CU's 160 KB: motions of up to 19 HR rows either way stay in the window
// (with 40 rows / 15, windows of the benchmark clip whose corners move by 16 to 21 rows cost 15 % more)
#ifndef SRX_FWD_RING
#define SRX_FWD_RING 47
#define SRX_FWD_MY 19
#endif
constexpr int kFwdRing = SRX_FWD_RING;
constexpr int kFwdMy = SRX_FWD_MY;
constexpr int kFwdMx = 16;
constexpr int kFwdQuads = kFwdStrip + 1 + kFwdMx / 2; // 16-byte groups per staged row
constexpr int kFwdRowBytes = 16 * kFwdQuads;          // one row of one plane
constexpr int kFwdSpan = 5 + 2 * kFwdMy;              // HR rows an LR row may read
constexpr int kFwdMaxWaves = 9;                       // window frames per block
static_assert(kFwdSpan + 4 <= kFwdRing && 2 * kFwdMy + 8 < kFwdRing,
              "rows staged during an LR row must not replace rows that row or the next one reads");
static_assert(kFwdMx % 4 == 0, "staged columns start at a multiple of four");

// One flag byte per (LR row, strip) of a frame's group map: kRowInside when every pixel the forward
// kernel gathers for that row of that strip lies inside its LDS window and no group is "far", so that
// the kernel may skip the checks; kRowOutside when a quarter of the lanes or more do not find their pixels
// there, so that the kernel goes to global memory right away (kRowMixed: a few lanes do not, e.g. those with
// a "far" group at a motion boundary - they alone go to global memory; these two kinds of rows are read in the
// long form of the entries, the others in the short form); kRowIdentity when every group of the row maps its four
// pixels onto themselves (zero motion: the base frame of a window, a static scene), so that the kernel
// needs no entry at all.  One wave per (strip, LR row); lanes as in the forward kernel.
constexpr uint8_t kRowMixed = 0, kRowInside = 1, kRowOutside = 2, kRowIdentity = 3;
// beside the kind (the low two bits): some lane of a mixed row has a source outside the window, i.e. the row makes
// a trip to global memory (a mixed row without it only needs the long form of the entries)
constexpr uint8_t kRowKindMask = 3, kRowLeaves = 4;
__global__ void k_group_flags(const uint32_t* __restrict__ grp, uint8_t* __restrict__ flags, int H, int W, int strips_x)
{
    const int w = W >> 2, h = H >> 2;
    const int strip = blockIdx.x, y = blockIdx.y, lane = threadIdx.x;
    const int x0 = strip * kFwdStrip;
    const int xl = min(max(x0 - 1 + lane, 0), w - 1);
    const int dc_lane = 4 * xl - (4 * (x0 - 1) - kFwdMx);
    const size_t entries = (size_t)h * w * 4;
    // ok: every group of the lane lies in the window by its short entry (then the long one says the same);
    // any: some group does by its long entry
    bool ok = true, any = false, every = true, same = true;
    for (int r = 0; r < 4; ++r)
    {
        const size_t e = ((size_t)y * w + xl) * 4 + r;
        const uint2 gs = make_uint2(grp[e], 0u), gl = make_uint2(grp[entries + e], grp[2 * entries + e]);
        same = same && gs.x == 0u;
        bool in_s, in_l;
        {
            const int dr = Grp<false>::dr(gs) + kFwdMy + 1 + r, dc = Grp<false>::dc(gs) + dc_lane;
            in_s = !Grp<false>::far(gs) && (unsigned)dr < (unsigned)kFwdSpan && (unsigned)dc < (unsigned)(4 * kFwdQuads);
            for (int j = 1; j < 4; ++j)
                in_s = in_s && (unsigned)(dr + Grp<false>::dy(gs, j)) < (unsigned)kFwdSpan &&
                       (unsigned)(dc + j + Grp<false>::dx(gs, j)) < (unsigned)(4 * kFwdQuads);
        }
        {
            const int dr = Grp<true>::dr(gl) + kFwdMy + 1 + r, dc = Grp<true>::dc(gl) + dc_lane;
            in_l = !Grp<true>::far(gl) && (unsigned)dr < (unsigned)kFwdSpan && (unsigned)dc < (unsigned)(4 * kFwdQuads);
            for (int j = 1; j < 4; ++j)
                in_l = in_l && (unsigned)(dr + Grp<true>::dy(gl, j)) < (unsigned)kFwdSpan &&
                       (unsigned)(dc + j + Grp<true>::dx(gl, j)) < (unsigned)(4 * kFwdQuads);
        }
        ok = ok && in_s;
        any = any || in_l;
        every = every && in_l;
    }
    const bool all_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
    const bool all_same = all_ok && __builtin_amdgcn_ballot_w64(!same) == 0;
    const int n_in = __builtin_popcountll(__builtin_amdgcn_ballot_w64(any));
    const bool leaves = __builtin_amdgcn_ballot_w64(!every) != 0;
    if (lane == 0)
        flags[(size_t)y * strips_x + strip] =
            all_same ? kRowIdentity
                     : (all_ok ? kRowInside : (n_in < 48 ? kRowOutside : (leaves ? kRowMixed | kRowLeaves : kRowMixed)));
}

// k_build_maps, k_group_maps and k_group_flags in one pass (scale 4): a thread works on one LR cell - the 4 x 4 HR
// pixels of rows 4y-1 .. 4y+2 and columns 4x .. 4x+3, i.e. the four groups of one 16-byte record of the group map -
// and a wave on one (LR row, strip) of the forward kernel, lanes as there, so that the row's kind falls out of
// two ballots at the end.  The bicubic taps of the 16 pixels come from 5 x 5 LR motions (a thread per HR pixel
// loads 16 for each): the row passes are shared by the pixels of a column, exactly the sums cubic_sample forms,
// in its order.  Everything leaves as 16-byte stores.  Block row h only finishes HR row H-1 (no group starts
// there).  The tap tables are cv::resize's (cubic_table_x / _y): per cell the four columns' first taps are
// (s, s, s + o, s + o) with o = 0 or 1, the four rows' likewise; a lane whose tables say otherwise (no geometry
// makes them) takes the per-pixel routine.
__global__ __launch_bounds__(64) void k_build_maps_cells(const float* __restrict__ rel, int h, int w, CubicTables t,
                                                         short2* __restrict__ fwd_map, short2* __restrict__ bwd_map,
                                                         uint32_t* __restrict__ dec_fwd, uint32_t* __restrict__ grp,
                                                         uint8_t* __restrict__ flags, int strips_x)
{
    const int H = 4 * h, W = 4 * w;
    const int strip = blockIdx.x, y = blockIdx.y, lane = threadIdx.x;
    const int x0 = strip * kFwdStrip;
    const int x = x0 - 1 + lane;
    const int xl = min(max(x, 0), w - 1);
    const bool owner = lane >= 1 && x < w;
    const bool last = y == h;
    const int n_rows = last ? 1 : 4;
    const int N0 = 4 * y - 1;
    int mrow[4], sy[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
    {
        mrow[r] = reflect101(min(N0 + r, H - 1), H);
        sy[r] = t.yofs[mrow[r]];
    }
    int base = sy[0];
#pragma unroll
    for (int r = 1; r < 4; ++r)
        base = r < n_rows ? min(base, sy[r]) : base;
    bool rows_ok = true;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        rows_ok = rows_ok && (r >= n_rows || sy[r] - base <= 1);
    const int X0 = 4 * xl;
    int sxv[4];
    float4 al[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
    {
        sxv[c] = t.xofs[X0 + c];
        al[c] = t.alpha[X0 + c];
    }
    const int sb = sxv[0];
    const bool shifted = sxv[2] != sb;
    const bool shared_taps = rows_ok && sxv[1] == sb && sxv[3] == sxv[2] && (sxv[2] == sb || sxv[2] == sb + 1);

    float mot[4][4][2]; // [HR row][column][x, y]
    if (shared_taps)
    {
        float hv[5][4][2];
#pragma unroll
        for (int i = 0; i < 5; ++i)
        {
            const float2* S = (const float2*)rel + (size_t)clampi(base - 1 + i, 0, h - 1) * w;
            float2 L[5];
#pragma unroll
            for (int k = 0; k < 5; ++k)
                L[k] = S[clampi(sb - 1 + k, 0, w - 1)];
#pragma unroll
            for (int comp = 0; comp < 2; ++comp)
            {
                float a[5], b[4];
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    a[k] = comp == 0 ? L[k].x : L[k].y;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    b[k] = shifted ? a[k + 1] : a[k];
                hv[i][0][comp] = a[0] * al[0].x + a[1] * al[0].y + a[2] * al[0].z + a[3] * al[0].w;
                hv[i][1][comp] = a[0] * al[1].x + a[1] * al[1].y + a[2] * al[1].z + a[3] * al[1].w;
                hv[i][2][comp] = b[0] * al[2].x + b[1] * al[2].y + b[2] * al[2].z + b[3] * al[2].w;
                hv[i][3][comp] = b[0] * al[3].x + b[1] * al[3].y + b[2] * al[3].z + b[3] * al[3].w;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const bool down = sy[r] != base; // the same for the whole block
            const float4 be = t.beta[mrow[r]];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int comp = 0; comp < 2; ++comp)
                {
                    const float v0 = down ? hv[1][c][comp] : hv[0][c][comp], v1 = down ? hv[2][c][comp] : hv[1][c][comp];
                    const float v2 = down ? hv[3][c][comp] : hv[2][c][comp], v3 = down ? hv[4][c][comp] : hv[3][c][comp];
                    mot[r][c][comp] = v0 * be.x + v1 * be.y + v2 * be.z + v3 * be.w;
                }
        }
    }
    else
    {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                float m[2];
                cubic_sample<2>(rel, h, w, sxv[c], al[c], sy[r], t.beta[mrow[r]], m);
                mot[r][c][0] = m[0], mot[r][c][1] = m[1];
            }
    }

    uint32_t g_short[4], g_long0[4], g_long1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
    {
        const int N = N0 + r; // the group's own row; its map row is mrow[r] (they differ for N = -1 only)
        int fx[4], fy[4], bx[4], by[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
        {
            const float mx = mot[r][c][0] * 4.0f, my = mot[r][c][1] * 4.0f;
            const float px = (float)(X0 + c), py = (float)mrow[r];
            fx[c] = round_sat16(px - mx), fy[c] = round_sat16(py - my);
            bx[c] = round_sat16(px + mx), by[c] = round_sat16(py + my);
        }
        if (owner && N >= 0 && r < n_rows)
        {
            const size_t o = (size_t)N * W + X0;
            typedef short __attribute__((ext_vector_type(8))) v8s;
            *(v8s*)(fwd_map + o) = v8s{(short)fx[0], (short)fy[0], (short)fx[1], (short)fy[1], (short)fx[2], (short)fy[2],
                                       (short)fx[3], (short)fy[3]};
            *(v8s*)(bwd_map + o) = v8s{(short)bx[0], (short)by[0], (short)bx[1], (short)by[1], (short)bx[2], (short)by[2],
                                       (short)bx[3], (short)by[3]};
            *(uint4*)(dec_fwd + o) = make_uint4(decode_fwd(fx[0], fy[0], H, W), decode_fwd(fx[1], fy[1], H, W),
                                                decode_fwd(fx[2], fy[2], H, W), decode_fwd(fx[3], fy[3], H, W));
        }
        // the group's entries in both forms (k_group_maps)
        const int col0 = bx[0], row0 = by[0];
        {
            const int ddx = clampi(col0 - X0, -63, 63), ddy = clampi(row0 - N, -63, 63);
            bool far = ddx != col0 - X0 || ddy != row0 - N;
            uint32_t others = 0;
#pragma unroll
            for (int i = 1; i <= 3; ++i)
            {
                const int dx = bx[i] - (col0 + i), dy = by[i] - row0;
                far = far || dx < -4 || dx > 3 || dy < -4 || dy > 3;
                others |= ((uint32_t)(dx & 7) | ((uint32_t)(dy & 7) << 3)) << (6 * (i - 1));
            }
            g_short[r] = far ? kGrpFar : (((uint32_t)ddx & 0x7Fu) | (((uint32_t)ddy & 0x7Fu) << 7) | (others << 14));
        }
        {
            const int ddx = clampi(col0 - X0, -16384, 16383), ddy = clampi(row0 - N, -16384, 16383);
            bool far = ddx != col0 - X0 || ddy != row0 - N;
            uint32_t w1 = 0;
#pragma unroll
            for (int i = 1; i <= 3; ++i)
            {
                const int dx = bx[i] - (col0 + i), dy = by[i] - row0;
                far = far || dx < -16 || dx > 15 || dy < -16 || dy > 15;
                w1 |= ((uint32_t)(dx & 31) | ((uint32_t)(dy & 31) << 5)) << (10 * (i - 1));
            }
            g_long0[r] = ((uint32_t)ddx & 0x7FFFu) | (((uint32_t)ddy & 0x7FFFu) << 15) | (far ? kGrpLongFar : 0u);
            g_long1[r] = w1;
        }
    }
    if (last)
        return;
    const size_t entries = (size_t)h * w * 4;
    if (owner)
    {
        const size_t cell = (size_t)y * w + xl;
        ((uint4*)grp)[cell] = make_uint4(g_short[0], g_short[1], g_short[2], g_short[3]);
        ((uint4*)(grp + entries))[cell] = make_uint4(g_long0[0], g_long0[1], g_long0[2], g_long0[3]);
        ((uint4*)(grp + 2 * entries))[cell] = make_uint4(g_long1[0], g_long1[1], g_long1[2], g_long1[3]);
    }
    // the row's kind (k_group_flags), from the entries just made
    const int dc_lane = 4 * xl - (4 * (x0 - 1) - kFwdMx);
    bool ok = true, any = false, every = true, same = true;
#pragma unroll
    for (int r = 0; r < 4; ++r)
    {
        const uint2 gs = make_uint2(g_short[r], 0u), gl = make_uint2(g_long0[r], g_long1[r]);
        same = same && gs.x == 0u;
        bool in_s, in_l;
        {
            const int dr = Grp<false>::dr(gs) + kFwdMy + 1 + r, dc = Grp<false>::dc(gs) + dc_lane;
            in_s = !Grp<false>::far(gs) && (unsigned)dr < (unsigned)kFwdSpan && (unsigned)dc < (unsigned)(4 * kFwdQuads);
#pragma unroll
            for (int j = 1; j < 4; ++j)
                in_s = in_s && (unsigned)(dr + Grp<false>::dy(gs, j)) < (unsigned)kFwdSpan &&
                       (unsigned)(dc + j + Grp<false>::dx(gs, j)) < (unsigned)(4 * kFwdQuads);
        }
        {
            const int dr = Grp<true>::dr(gl) + kFwdMy + 1 + r, dc = Grp<true>::dc(gl) + dc_lane;
            in_l = !Grp<true>::far(gl) && (unsigned)dr < (unsigned)kFwdSpan && (unsigned)dc < (unsigned)(4 * kFwdQuads);
#pragma unroll
            for (int j = 1; j < 4; ++j)
                in_l = in_l && (unsigned)(dr + Grp<true>::dy(gl, j)) < (unsigned)kFwdSpan &&
                       (unsigned)(dc + j + Grp<true>::dx(gl, j)) < (unsigned)(4 * kFwdQuads);
        }
        ok = ok && in_s;
        any = any || in_l;
        every = every && in_l;
    }
    const bool all_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
    const bool all_same = all_ok && __builtin_amdgcn_ballot_w64(!same) == 0;
    const int n_in = __builtin_popcountll(__builtin_amdgcn_ballot_w64(any));
    const bool leaves = __builtin_amdgcn_ballot_w64(!every) != 0;
    if (lane == 0)
        flags[(size_t)y * strips_x + strip] =
            all_same ? kRowIdentity
                     : (all_ok ? kRowInside : (n_in < 48 ? kRowOutside : (leaves ? kRowMixed | kRowLeaves : kRowMixed)));
}

// ps == 0: interleaved destination; otherwise planar with plane stride ps
template <int CH>
__global__ void k_resize_cubic(const float* __restrict__ src, int srows, int scols, CubicTables t,
                               float* __restrict__ dst, int drows, int dcols, size_t ps)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dcols || y >= drows)
        return;
    float v[CH];
    cubic_sample<CH>(src, srows, scols, t.xofs[x], t.alpha[x], t.yofs[y], t.beta[y], v);
    const size_t o = (size_t)y * dcols + x;
#pragma unroll
    for (int c = 0; c < CH; ++c)
        dst[ps ? c * ps + o : o * CH + c] = v[c];
}

// The same enlargement by 4 into planes, a thread per LR cell (the 4 x 4 HR pixels of rows 4y .. 4y+3 and columns
// 4x .. 4x+3): their taps come from 5 x 5 LR pixels, the row passes are shared by the pixels of a column - exactly
// the sums cubic_sample forms, in its order - and every plane's four pixels of a row leave as one 16-byte store.
// Per cell the four columns' first taps are (s, s, s + o, s + o) with o = 0 or 1, the four rows' likewise
// (cubic_table_x / _y at scale 4); a cell whose tables say otherwise takes the per-pixel routine.
template <int CH>
__global__ __launch_bounds__(256) void k_resize_cubic_cells(const float* __restrict__ src, int srows, int scols,
                                                            CubicTables t, float* __restrict__ dst, size_t ps)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= scols || y >= srows)
        return;
    const int dcols = 4 * scols;
    int sy[4], sxv[4];
    float4 al[4], be[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
    {
        sy[r] = t.yofs[4 * y + r], be[r] = t.beta[4 * y + r];
        sxv[r] = t.xofs[4 * x + r], al[r] = t.alpha[4 * x + r];
    }
    const int rb = sy[0], sb = sxv[0];
    const bool down = sy[2] != rb, shifted = sxv[2] != sb;
    const bool shared_taps = sy[1] == rb && sy[3] == sy[2] && (sy[2] == rb || sy[2] == rb + 1) && sxv[1] == sb &&
                             sxv[3] == sxv[2] && (sxv[2] == sb || sxv[2] == sb + 1);
    float out[4][4][CH]; // [HR row][column][channel]
    if (shared_taps)
    {
        float hv[5][4][CH];
#pragma unroll
        for (int i = 0; i < 5; ++i)
        {
            const float* S = src + (size_t)clampi(rb - 1 + i, 0, srows - 1) * scols * CH;
            int col[5];
#pragma unroll
            for (int k = 0; k < 5; ++k)
                col[k] = clampi(sb - 1 + k, 0, scols - 1) * CH;
#pragma unroll
            for (int c = 0; c < CH; ++c)
            {
                float a[5], b[4];
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    a[k] = S[col[k] + c];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    b[k] = shifted ? a[k + 1] : a[k];
                hv[i][0][c] = a[0] * al[0].x + a[1] * al[0].y + a[2] * al[0].z + a[3] * al[0].w;
                hv[i][1][c] = a[0] * al[1].x + a[1] * al[1].y + a[2] * al[1].z + a[3] * al[1].w;
                hv[i][2][c] = b[0] * al[2].x + b[1] * al[2].y + b[2] * al[2].z + b[3] * al[2].w;
                hv[i][3][c] = b[0] * al[3].x + b[1] * al[3].y + b[2] * al[3].z + b[3] * al[3].w;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const bool dn = r >= 2 && down; // rows 0, 1 start at rb, rows 2, 3 at rb or rb + 1
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < CH; ++c)
                {
                    const float v0 = dn ? hv[1][q][c] : hv[0][q][c], v1 = dn ? hv[2][q][c] : hv[1][q][c];
                    const float v2 = dn ? hv[3][q][c] : hv[2][q][c], v3 = dn ? hv[4][q][c] : hv[3][q][c];
                    out[r][q][c] = v0 * be[r].x + v1 * be[r].y + v2 * be[r].z + v3 * be[r].w;
                }
        }
    }
    else
    {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                cubic_sample<CH>(src, srows, scols, sxv[q], al[q], sy[r], be[r], out[r][q]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < CH; ++c)
            *(float4*)(dst + c * ps + (size_t)(4 * y + r) * dcols + 4 * x) =
                make_float4(out[r][0][c], out[r][1][c], out[r][2][c], out[r][3][c]);
}

// interleaved <-> planar copies (stage-level entry points and x_full output)
template <int CH>
__global__ void k_relayout(const float* __restrict__ src, float* __restrict__ dst, size_t count, size_t ps,
                           int to_planar)
{
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= count)
        return;
#pragma unroll
    for (int c = 0; c < CH; ++c)
    {
        if (to_planar)
            dst[c * ps + o] = src[o * CH + c];
        else
            dst[o * CH + c] = src[c * ps + o];
    }
}

// ===========================================================================
// Generic kernels
// ===========================================================================

// BTV regulariser at one pixel (lib/src/btv_l1.cpp:212-235), straight from
// global memory.  Border band of width `radius` stays zero.
template <int CH>
__device__ __forceinline__ void btv_at(const float* __restrict__ X, int H, int W, int i, int j, const BtvW& w,
                                       float (&reg)[CH])
{
#pragma unroll
    for (int c = 0; c < CH; ++c)
        reg[c] = 0.0f;
    const int p = w.radius;
    if (i < p || i >= H - p || j < p || j >= W - p)
        return;
    float v[CH];
    const float* P = X + ((size_t)i * W + j) * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c)
        v[c] = P[c];
    int ind = 0;
    for (int m = 0; m <= p; ++m)
        for (int l = p; l + m >= 0; --l, ++ind)
        {
            const float* P3 = X + ((size_t)(i + m) * W + (j + l)) * CH;
            const float* P2 = X + ((size_t)(i - m) * W + (j - l)) * CH;
            const float wt = w.w[ind];
#pragma unroll
            for (int c = 0; c < CH; ++c)
                reg[c] += wt * (dsign(v[c], P3[c]) - dsign(P2[c], v[c]));
        }
}

template <int CH>
__global__ void k_btv_regularization(const float* __restrict__ X, int H, int W, BtvW w, float* __restrict__ dst)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (j >= W || i >= H)
        return;
    float reg[CH];
    btv_at<CH>(X, H, W, i, j, w, reg);
    float* D = dst + ((size_t)i * W + j) * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c)
        D[c] = reg[c];
}

// a(Y, X) = Xhr(bwd(Y, X)) or 0 outside   (remap NEAREST, constant border)
template <int CH>
__device__ __forceinline__ void fetch_warped(const float* __restrict__ X, int H, int W, size_t ps,
                                             const short2* __restrict__ map, int Yr, int Xr, float (&o)[CH])
{
    const short2 m = map[(size_t)Yr * W + Xr];
    if ((unsigned)(int)m.x < (unsigned)W && (unsigned)(int)m.y < (unsigned)H)
    {
        const size_t pi = (size_t)m.y * W + m.x;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            o[c] = X[ps ? c * ps + pi : pi * CH + c];
    }
    else
    {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            o[c] = 0.0f;
    }
}

// Row pass of the separable Gaussian at (Yr, Xc) in OpenCV's operation order:
// n <= 5: centre*k0 + (l1+r1)*k1 + (l2+r2)*k2; n >= 7: taps left to right.
template <int CH>
__device__ __forceinline__ void warped_blur_row(const float* __restrict__ X, int H, int W, size_t ps,
                                                const short2* __restrict__ map, int Yr, int Xc, const GaussK& g,
                                                float (&r)[CH])
{
    const int n = g.n, hh = n >> 1;
    float t[CH], u[CH];
    if (n <= 5)
    {
        fetch_warped<CH>(X, H, W, ps, map, Yr, Xc, t);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            r[c] = t[c] * g.k[hh];
        for (int j = 1; j <= hh; ++j)
        {
            fetch_warped<CH>(X, H, W, ps, map, Yr, reflect101(Xc - j, W), t);
            fetch_warped<CH>(X, H, W, ps, map, Yr, reflect101(Xc + j, W), u);
#pragma unroll
            for (int c = 0; c < CH; ++c)
                r[c] += (t[c] + u[c]) * g.k[hh + j];
        }
    }
    else
    {
        fetch_warped<CH>(X, H, W, ps, map, Yr, reflect101(Xc - hh, W), t);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            r[c] = g.k[0] * t[c];
        for (int j = 1; j < n; ++j)
        {
            fetch_warped<CH>(X, H, W, ps, map, Yr, reflect101(Xc - hh + j, W), t);
#pragma unroll
            for (int c = 0; c < CH; ++c)
                r[c] += g.k[j] * t[c];
        }
    }
}

// sign(y - D H M X) at one LR pixel, any scale / blur size (lib/src/btv_l1.cpp:428-434)
template <int CH>
__device__ __forceinline__ uint32_t forward_sign_generic(const float* __restrict__ X, int H, int W, size_t ps,
                                                         const short2* __restrict__ map,
                                                         const float* __restrict__ yk, int x, int y, int scale,
                                                         const GaussK& g)
{
    const int w = W / scale;
    const int Yc = y * scale, Xc = x * scale;
    const int hh = g.n >> 1;
    float acc[CH], ra[CH], rb[CH];
    // column pass: centre row first, then symmetric pairs outwards
    warped_blur_row<CH>(X, H, W, ps, map, Yc, Xc, g, ra);
#pragma unroll
    for (int c = 0; c < CH; ++c)
        acc[c] = ra[c] * g.k[hh];
    for (int j = 1; j <= hh; ++j)
    {
        warped_blur_row<CH>(X, H, W, ps, map, reflect101(Yc - j, H), Xc, g, ra);
        warped_blur_row<CH>(X, H, W, ps, map, reflect101(Yc + j, H), Xc, g, rb);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            acc[c] += (ra[c] + rb[c]) * g.k[hh + j];
    }
    const float* yp = yk + ((size_t)y * w + x) * CH;
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c)
        s |= (uint32_t)((int)dsign(yp[c], acc[c]) + 1) << (10 * c);
    return s;
}

// ps == 0: interleaved X; otherwise planar with plane stride ps
template <int CH>
__global__ void k_forward_sign(const float* __restrict__ X, int H, int W, size_t ps, PtrList bwd_maps, PtrList ys,
                               MutPtrList signs, int scale, GaussK g)
{
    const int h = H / scale, w = W / scale;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = blockIdx.z;
    if (x >= w || y >= h)
        return;
    ((uint32_t*)signs.p[k])[(size_t)y * w + x] =
        forward_sign_generic<CH>(X, H, W, ps, (const short2*)bwd_maps.p[k], (const float*)ys.p[k], x, y, scale, g);
}

// The same values by tiles, for any scale and blur size: a block warps the HR region that a tile of
// 32 x 8 LR pixels of one frame needs into LDS once (one gather per HR pixel instead of one per tap),
// runs the row pass at the decimated columns, then the column pass at the decimated rows.
constexpr int kTileLrW = 32, kTileLrH = 8;

inline size_t forward_tile_lds_bytes(int ch, int scale, int n)
{
    const size_t hr_r = (size_t)(kTileLrH - 1) * scale + n, hr_c = (size_t)(kTileLrW - 1) * scale + n;
    return (hr_r * hr_c + hr_r * kTileLrW) * ch * sizeof(float);
}

// N5: the blur has five taps (loops unrolled); otherwise any size
template <int CH, bool N5>
__global__ __launch_bounds__(256) void k_forward_sign_tile(const float* __restrict__ X, int H, int W, size_t ps,
                                                           PtrList bwd_maps, PtrList ys, MutPtrList signs, int scale,
                                                           GaussK g)
{
    extern __shared__ __align__(16) float tile_lds[];
    const int h = H / scale, w = W / scale;
    const int n = N5 ? 5 : g.n, hh = n >> 1;
    const int hr_r = (kTileLrH - 1) * scale + n, hr_c = (kTileLrW - 1) * scale + n;
    const int n_px = hr_r * hr_c;
    float* A = tile_lds;                // [CH][hr_r][hr_c] warped pixels
    float* R = tile_lds + CH * n_px;    // [CH][hr_r][kTileLrW] row-pass results
    const int k = blockIdx.z;
    const int x0 = blockIdx.x * kTileLrW, y0 = blockIdx.y * kTileLrH;
    const int X0 = x0 * scale - hh, Y0 = y0 * scale - hh;
    const short2* map = (const short2*)bwd_maps.p[k];
    const int tid = threadIdx.x;

    // warp: pixel idx = tid, tid + 256, ... of the region; (r, c) advance without divisions
    const bool interior = X0 >= 0 && Y0 >= 0 && X0 + hr_c <= W && Y0 + hr_r <= H; // no reflection in this tile
    const int r_step = 256 / hr_c, c_step = 256 - r_step * hr_c;
    int r = tid / hr_c, c = tid - r * hr_c;
    for (int idx = tid; idx < n_px; idx += 256)
    {
        const int Yr = interior ? Y0 + r : reflect101(Y0 + r, H), Xr = interior ? X0 + c : reflect101(X0 + c, W);
        const short2 m = map[(unsigned)(Yr * W + Xr)];
        const bool in = (unsigned)(int)m.x < (unsigned)W && (unsigned)(int)m.y < (unsigned)H;
        const unsigned pi = in ? (unsigned)((int)m.y * W + (int)m.x) : 0u;
#pragma unroll
        for (int ch = 0; ch < CH; ++ch)
        {
            const float v = X[ps ? ch * ps + pi : (size_t)pi * CH + ch];
            A[ch * n_px + idx] = in ? v : 0.0f; // remap's constant border
        }
        r += r_step, c += c_step;
        if (c >= hr_c)
            c -= hr_c, ++r;
    }
    __syncthreads();
    // row pass in OpenCV's operation order (see warped_blur_row)
    for (int idx = tid; idx < hr_r * kTileLrW; idx += 256)
    {
        const int rr_ = idx / kTileLrW, xl = idx - rr_ * kTileLrW;
        const int cc = xl * scale + hh;
#pragma unroll
        for (int ch = 0; ch < CH; ++ch)
        {
            const float* a = A + ch * n_px + rr_ * hr_c;
            float s;
            if (N5)
            {
                s = a[cc] * g.k[2];
                s += (a[cc - 1] + a[cc + 1]) * g.k[3];
                s += (a[cc - 2] + a[cc + 2]) * g.k[4];
            }
            else if (n <= 5)
            {
                s = a[cc] * g.k[hh];
                for (int j = 1; j <= hh; ++j)
                    s += (a[cc - j] + a[cc + j]) * g.k[hh + j];
            }
            else
            {
                s = g.k[0] * a[cc - hh];
                for (int j = 1; j < n; ++j)
                    s += g.k[j] * a[cc - hh + j];
            }
            R[(ch * hr_r + rr_) * kTileLrW + xl] = s;
        }
    }
    __syncthreads();
    const int xl = tid & (kTileLrW - 1), yl = tid / kTileLrW;
    const int x = x0 + xl, y = y0 + yl;
    if (x >= w || y >= h)
        return;
    const int rc = yl * scale + hh;
    const float* yp = (const float*)ys.p[k] + ((size_t)y * w + x) * CH;
    uint32_t sg = 0;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
    {
        const float* q = R + ch * hr_r * kTileLrW + xl;
        float acc;
        if (N5)
        {
            acc = q[rc * kTileLrW] * g.k[2];
            acc += (q[(rc - 1) * kTileLrW] + q[(rc + 1) * kTileLrW]) * g.k[3];
            acc += (q[(rc - 2) * kTileLrW] + q[(rc + 2) * kTileLrW]) * g.k[4];
        }
        else
        {
            acc = q[rc * kTileLrW] * g.k[hh];
            for (int j = 1; j <= hh; ++j)
                acc += (q[(rc - j) * kTileLrW] + q[(rc + j) * kTileLrW]) * g.k[hh + j];
        }
        sg |= (uint32_t)((int)dsign(yp[ch], acc) + 1) << (10 * ch);
    }
    ((uint32_t*)signs.p[k])[(size_t)y * w + x] = sg;
}

// Transposed chain.  B(fy, fx) = gauss(zero_insert(sign))(fy, fx) evaluated
// lazily: the zero-inserted image is non-zero only where both coordinates are
// multiples of scale.
// SS = words per cell: 1 for a plain plane, 2 for a pair plane
// v % scale == 0 and v / scale for v >= 0 without an integer division when scale is a power of two
__device__ __forceinline__ bool on_lr_grid(int v, int scale)
{
    return (scale & (scale - 1)) == 0 ? (v & (scale - 1)) == 0 : v % scale == 0;
}
__device__ __forceinline__ int to_lr(int v, int scale)
{
    return (scale & (scale - 1)) == 0 ? v >> (31 - __clz(scale)) : v / scale;
}

template <int CH, int SS = 1>
__device__ __forceinline__ void sign_at(const uint32_t* __restrict__ sg, int w, int scale, int Yr, int Xr,
                                        float (&o)[CH])
{
    // caller guarantees Yr % scale == 0
    if (on_lr_grid(Xr, scale))
    {
        const uint32_t s = sg[((size_t)to_lr(Yr, scale) * w + to_lr(Xr, scale)) * SS];
#pragma unroll
        for (int c = 0; c < CH; ++c)
            o[c] = (float)((int)((s >> (10 * c)) & 3u) - 1);
    }
    else
    {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            o[c] = 0.0f;
    }
}

template <int CH, int SS = 1>
__device__ __forceinline__ void upsampled_blur_row(const uint32_t* __restrict__ sg, int W, int scale, int Yr, int fx,
                                                   const GaussK& g, float (&r)[CH])
{
    if (!on_lr_grid(Yr, scale))
    {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            r[c] = 0.0f;
        return;
    }
    const int w = to_lr(W, scale);
    const int n = g.n, hh = n >> 1;
    float t[CH], u[CH];
    if (n <= 5)
    {
        sign_at<CH, SS>(sg, w, scale, Yr, fx, t);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            r[c] = t[c] * g.k[hh];
        for (int j = 1; j <= hh; ++j)
        {
            sign_at<CH, SS>(sg, w, scale, Yr, reflect101(fx - j, W), t);
            sign_at<CH, SS>(sg, w, scale, Yr, reflect101(fx + j, W), u);
#pragma unroll
            for (int c = 0; c < CH; ++c)
                r[c] += (t[c] + u[c]) * g.k[hh + j];
        }
    }
    else
    {
        sign_at<CH, SS>(sg, w, scale, Yr, reflect101(fx - hh, W), t);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            r[c] = g.k[0] * t[c];
        for (int j = 1; j < n; ++j)
        {
            sign_at<CH, SS>(sg, w, scale, Yr, reflect101(fx - hh + j, W), t);
#pragma unroll
            for (int c = 0; c < CH; ++c)
                r[c] += g.k[j] * t[c];
        }
    }
}

// (M^T H^T D^T sign)(i, j) for one frame: gather at the forward-map target
template <int CH, int SS = 1>
__device__ __forceinline__ void backproject_generic(const uint32_t* __restrict__ sg, int H, int W, int scale, int fx,
                                                    int fy, const GaussK& g, float (&a)[CH])
{
    const int hh = g.n >> 1;
    float ra[CH], rb[CH];
    upsampled_blur_row<CH, SS>(sg, W, scale, fy, fx, g, ra);
#pragma unroll
    for (int c = 0; c < CH; ++c)
        a[c] = ra[c] * g.k[hh];
    for (int t = 1; t <= hh; ++t)
    {
        upsampled_blur_row<CH, SS>(sg, W, scale, reflect101(fy - t, H), fx, g, ra);
        upsampled_blur_row<CH, SS>(sg, W, scale, reflect101(fy + t, H), fx, g, rb);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            a[c] += (ra[c] + rb[c]) * g.k[hh + t];
    }
}

// X_out = X + tau * (sum_k M_k^T H^T D^T sign_k - lambda * btv(X))
// lib/src/btv_l1.cpp:437-452; both addWeighted steps evaluate in double.
template <int CH>
__global__ void k_backproject_update(const float* __restrict__ X, float* __restrict__ X_out, int H, int W,
                                     PtrList fwd_maps, PtrList signs, int K, int scale, GaussK g, BtvW bw, double tau,
                                     double lambda)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (j >= W || i >= H)
        return;
    float diff[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
        diff[c] = 0.0f;
    for (int k = 0; k < K; ++k)
    {
        const short2 m = ((const short2*)fwd_maps.p[k])[(size_t)i * W + j];
        const int fx = m.x, fy = m.y;
        if ((unsigned)fx >= (unsigned)W || (unsigned)fy >= (unsigned)H)
            continue; // remap border: constant 0
        float a[CH];
        backproject_generic<CH>((const uint32_t*)signs.p[k], H, W, scale, fx, fy, g, a);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            diff[c] = diff[c] + a[c];
    }
    const float* P = X + ((size_t)i * W + j) * CH;
    float* D = X_out + ((size_t)i * W + j) * CH;
    if (lambda > 0)
    {
        float reg[CH];
        btv_at<CH>(X, H, W, i, j, bw, reg);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            diff[c] = (float)((double)diff[c] + (double)reg[c] * (-lambda));
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
        D[c] = (float)((double)P[c] + (double)diff[c] * tau);
}

// ===========================================================================
// Specialised kernels: scale 4, 5-tap blur (and BTV kernel size 7 for the update)
// ===========================================================================

// value held by lane-1 of the wave (0 in lane 0)
__device__ __forceinline__ float from_left_lane(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
}

// four floats at any 4-byte aligned address: one global_load_dwordx4
struct __attribute__((packed, aligned(4))) float4_a4
{
    float x, y, z, w;
};
// N floats fetched by one load instruction
template <int N> struct __attribute__((packed, aligned(4))) packed_floats
{
    float v[N];
};

// Forward chain sign(y_k - D H M_k X) with the gathers served from LDS.
//
// One block per tile (strip of 63 LR columns x a range of LR rows), one wave per window frame.
// Every frame of the window gathers from nearly the same region of X - the motions are a few
// pixels - so the block keeps that region in LDS and X leaves the caches once per tile instead of
// once per frame: a ring of kFwdRing HR rows x (strip + kFwdMx columns either side), all planes,
// walks down the tile, four new HR rows per LR row, staged by all waves with aligned 16-byte loads
// (zeros around the image: cv::remap's constant border).  An LR row may read the HR rows
// 4y-2-kFwdMy .. 4y+2+kFwdMy; a lane with a source outside that window, or with a "far" group,
// re-reads its four pixels from global memory, so the window only decides speed, never values.
//
// LDS layout of one staged row and plane: four sub-arrays, one per column phase (column & 3), of
// kFwdQuads floats each.  Neighbouring lanes read pixels four columns apart, i.e. neighbouring
// elements of one sub-array: no bank conflicts, whatever the alignment of the motion.
//
// Lane l works on LR column x = x0 - 1 + l and holds, per HR row and plane, the warped pixels of HR
// columns 4x .. 4x+3.  The 5-tap row pass centred on column 4x is
//     a0*k0 + (L.a3 + a1)*k1 + (L.a2 + a2)*k2
// with L the lane to the left (lane 0 only serves as lane 1's left neighbour; at the image's left
// edge it hands over columns 1 and 2, where reflect101 sends columns -1 and -2).  The wave walks down its LR rows keeping the five
// row-pass results of the current LR row in registers (four new HR rows per LR row).  Signs go to
// the pair plane (layout at the top of this file).
// The ring is re-used with one barrier per LR row: rows staged during LR row y replace rows that
// neither row y nor row y+1 reads (static_assert below).
#ifndef SRX_A2_WAVES
#define SRX_A2_WAVES 3
#endif

// Tiles of the forward kernel cut by what their rows cost.  A tile's LR rows move at the pace of the slowest wave
// (one barrier per LR row), and a row with gathers that leave the LDS window (a few lanes: kRowMixed with
// kRowLeaves, most lanes: kRowOutside) goes to global memory for them and takes longer; with an even cut the few tiles that hold such
// rows (image corners under a rotation, a border under a pan) finish long after the others, and the launch - one
// tile per CU - ends with them.  Here every (LR row, strip) gets a cost: that of a row served from the window plus
// a surcharge per frame whose wave goes to global memory there (such waves share the CU's path to the caches, so
// the row's time grows with their number); strips get tiles in proportion to their cost (bisection on the cost per tile, a tile's start-up
// included), and a strip's rows are cut where the running cost passes the next multiple of the strip's share.
// Which rows a tile holds never changes a value.  One block, a wave per strip; plan[t] = (strip, first row, end
// row, 0); spare entries and tiles that a single costly row left without rows have first row == end row.
constexpr int kPlanThreads = 1024, kPlanMaxStrips = 256;
constexpr int kPlanLdsCosts = 12288; // (LR rows x strips) up to which the costs are kept in LDS
constexpr int kPlanPerTile = 80; // a tile's start-up in units of PlanCost (five rows served from the window)
struct PlanCost
{
    int inside, mixed, outside, most_extra, min_rows;
};
__global__ __launch_bounds__(kPlanThreads) void k_plan_tiles(PtrList grp_bwd, int K, int H, int W, int strips_x,
                                                             int n_tiles, PlanCost pc, int* __restrict__ cost,
                                                             int4* __restrict__ plan)
{
    __shared__ int s_total[kPlanMaxStrips], s_first[kPlanMaxStrips + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = H >> 2;
    const size_t flag_at = 3 * ((size_t)H * W / 4) * sizeof(uint32_t);
    // (four rows' flags per load, the loads of all frames in flight together; the costs stay in LDS where they fit)
    __shared__ int s_cost[kPlanLdsCosts];
    const int cells = h * strips_x;
    if (cells <= kPlanLdsCosts)
        cost = s_cost;
    for (int i4 = tid; 4 * i4 < cells; i4 += kPlanThreads)
    {
        int extra[4] = {0, 0, 0, 0};
        for (int k0 = 0; k0 < K; k0 += 8) // eight frames' loads in flight together
        {
            uint32_t f4[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
            {
                f4[u] = 0; // (flag_at is a multiple of 4; a word's last bytes may belong to nothing: never used)
                if (k0 + u < K)
                    f4[u] = ((const uint32_t*)((const uint8_t*)grp_bwd.p[k0 + u] + flag_at))[i4];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                {
                    const uint32_t flag = (f4[u] >> (8 * b)) & 0xFFu; // 0 (a mixed row that stays inside) beyond K
                    extra[b] += (flag & kRowKindMask) == kRowOutside ? pc.outside : ((flag & kRowLeaves) ? pc.mixed : 0);
                }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (4 * i4 + b < cells)
                cost[4 * i4 + b] = pc.inside + min(extra[b], pc.most_extra);
    }
    for (int t = tid; t < n_tiles; t += kPlanThreads)
        plan[t] = make_int4(0, 0, 0, 0);
    __syncthreads();
    // a lane owns `chunk` neighbouring rows of its wave's strip
    const int chunk = (h + 63) / 64;
    const int ya = min(lane * chunk, h), yb = min(ya + chunk, h);
    for (int sidx = wave; sidx < strips_x; sidx += kPlanThreads / 64)
    {
        int part = 0;
        for (int y = ya; y < yb; ++y)
            part += cost[y * strips_x + sidx];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            part += __shfl_xor(part, o, 64);
        if (lane == 0)
            s_total[sidx] = part;
    }
    __syncthreads();
    // the smallest cost per tile L (start-up included) at which the strips need no more than n_tiles tiles:
    // tiles(s, L) = ceil(total_s / (L - kPlanPerTile)), at least 1 and at most h / min_rows.  One wave searches,
    // its lanes taking the strips in turn (every thread on its own would spend the kernel's time on the divisions).
    const int most = max(1, h / max(pc.min_rows, 1));
    if (wave == 0)
    {
        int top = 0;
        for (int sidx = lane; sidx < strips_x; sidx += 64)
            top = max(top, s_total[sidx]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            top = max(top, __shfl_xor(top, o, 64));
        int lo = kPlanPerTile + 1, hi = kPlanPerTile + max(top, 1);
        while (lo < hi) // lo, hi and mid are the same in every lane
        {
            const int mid = lo + (hi - lo) / 2;
            int need = 0;
            for (int sidx = lane; sidx < strips_x; sidx += 64)
                need += min(most, max(1, (s_total[sidx] + (mid - kPlanPerTile) - 1) / (mid - kPlanPerTile)));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                need += __shfl_xor(need, o, 64);
            if (need <= n_tiles)
                hi = mid;
            else
                lo = mid + 1;
        }
        const int share = lo - kPlanPerTile;
        // first tile of every strip: a running sum over the strips, 64 at a time
        int base = 0;
        for (int s0 = 0; s0 < strips_x; s0 += 64)
        {
            const int sidx = s0 + lane;
            const int mine = sidx < strips_x ? min(most, max(1, (s_total[sidx] + share - 1) / share)) : 0;
            int incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1)
            {
                const int v = __shfl_up(incl, o, 64);
                if (lane >= o)
                    incl += v;
            }
            if (sidx < strips_x)
                s_first[sidx] = base + incl - mine;
            base += __shfl(incl, 63, 64);
        }
        if (lane == 0)
            s_first[strips_x] = base; // <= n_tiles
    }
    __syncthreads();
    int* const fields = (int*)plan;
    for (int sidx = wave; sidx < strips_x; sidx += kPlanThreads / 64)
    {
        const int first = s_first[sidx], T = s_first[sidx + 1] - first;
        const long long total = max(s_total[sidx], 1);
        for (int j = lane; j < T; j += 64)
            fields[4 * (first + j)] = sidx;
        if (lane == 0)
        {
            fields[4 * first + 1] = 0;
            fields[4 * (first + T - 1) + 2] = h;
        }
        // running cost in front of the lane's rows
        int part = 0;
        for (int y = ya; y < yb; ++y)
            part += cost[y * strips_x + sidx];
        int before = part;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
        {
            const int v = __shfl_up(before, o, 64);
            if (lane >= o)
                before += v;
        }
        long long acc = before - part;
        // tile j of the strip starts behind the first row at which the running cost reaches j shares, i.e.
        // acc * T >= j * total (one division per lane for the shares reached in front of its rows, products after)
        int j = (int)(acc * T / total);
        for (int y = ya; y < yb; ++y)
        {
            acc += cost[y * strips_x + sidx];
            while (j + 1 < T && (long long)(j + 1) * total <= acc * T)
            {
                ++j;
                fields[4 * (first + j) + 1] = y + 1;
                fields[4 * (first + j - 1) + 2] = y + 1;
            }
        }
    }
}

inline size_t forward_lds_bytes(int ch)
{
    return (size_t)kFwdRing * ch * kFwdRowBytes;
}

// Y8: the LR frames are the 8-bit originals (a quarter of the bytes of their float copies; the values are the same)
template <int CH, bool Y8>
__global__ __launch_bounds__(64 * kFwdMaxWaves, SRX_A2_WAVES) void k_forward_sign_lds(
    const float* __restrict__ X, int H, int W, size_t ps, PtrList raw_bwd, PtrList grp_bwd, PtrList ys, MutPtrList signs,
    int K, GaussK g, int strips_x, int rows_per_tile, unsigned n_tiles, int frame_shift, const int4* __restrict__ plan)
{
    extern __shared__ __align__(16) char ring[];
    constexpr int kSlotBytes = CH * kFwdRowBytes;
    constexpr int kItemsPerRow = CH * kFwdQuads;
    constexpr int kRingBytes = kFwdRing * kSlotBytes;

#ifdef SRX_STAMPS
    const unsigned long long t_begin = SRX_TICK();
#endif
    const unsigned tile = xcd_contiguous(blockIdx.x, n_tiles);
    if (tile >= n_tiles)
        return;
    if ((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring != 0u)
        __builtin_trap(); // the gathers below address the ring by offset
    // the tile's strip and rows: an even cut of the rows, or what k_plan_tiles laid down
    int strip = tile % strips_x, tile_y0 = (tile / strips_x) * rows_per_tile, tile_y1 = tile_y0 + rows_per_tile;
    if (plan)
    {
        const int4 t = plan[tile];
        strip = t.x, tile_y0 = t.y, tile_y1 = t.z;
        if (tile_y0 >= tile_y1)
            return; // a spare entry (the whole block leaves before any barrier)
    }
    const int h = H >> 2, w = W >> 2;
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int kw = blockIdx.y * (nthreads >> 6) + (tid >> 6);
    const bool active = kw < K;
    // a spare wave shadows frame 0 (it still stages and meets the barriers); wave-uniform, so that
    // everything derived from the frame's pointers stays in scalar registers
    // Wave kw works on frame (kw + frame_shift) mod K: the launcher picks the shift so that a frame with zero
    // motion (the cheapest one: no map entries, no address arithmetic) goes to the last wave, which in a block
    // of 4n+1 waves sits on the SIMD that carries a wave more than the others.
    const int k = __builtin_amdgcn_readfirstlane(active ? (kw + frame_shift >= K ? kw + frame_shift - K : kw + frame_shift) : 0);
    const int x0 = strip * kFwdStrip, y0 = tile_y0;
    const int y1 = min(tile_y1, h);

    const int x = x0 - 1 + lane;          // LR column of this lane
    const int xl = min(max(x, 0), w - 1); // column it loads (clamped copies are never stored)
    const bool owner = active && lane >= 1 && x < w;
    // At the image's left edge the row pass of LR column 0 needs columns -1 and -2 of the warped
    // image, which reflect101 maps to columns 1 and 2.  Lane 0 of the first strip (x = -1, loading
    // column 0's group like lane 1) therefore delivers the pixel of column 1 in the place of column
    // 3: its neighbour then finds columns 1 and 2 where it looks for columns -1 and -2.
    const bool mirror = x == -1;

    const int c0 = 4 * (x0 - 1) - kFwdMx;  // HR column of element 0 of the staged rows
    const int rbase = 4 * y0 - 2 - kFwdMy; // HR row in ring slot 0
    const int dc_lane = 4 * xl - c0;

    // short entries, long entries word 0 and word 1: three planes of h*w 16-byte records
    const uint4* grp_base = (const uint4*)grp_bwd.p[k];
    const uint4* grp0 = grp_base + xl;
    // per LR row: 1 = every gather of this strip's row stays inside the window (k_group_flags)
    const uint8_t* flags = (const uint8_t*)((const uint32_t*)grp_bwd.p[k] + 3 * ((size_t)H * W / 4)) + strip;
    typedef typename std::conditional<Y8, uint8_t, float>::type y_elem;
    struct __attribute__((packed)) y_pack
    {
        y_elem v[CH];
    };
    const y_elem* yk = (const y_elem*)ys.p[k] + (size_t)xl * CH;
    uint32_t* sp = (uint32_t*)signs.p[k] + 2 * (size_t)xl;
    const float k0 = g.k[2], k1 = g.k[3], k2 = g.k[4];

    // ---- staging -----------------------------------------------------------------------------
    // piece -> (row within the batch, plane, 16-byte group)
    auto stage_piece = [&](int item, int row_first, int slot_first) {
        const int rr_ = item / kItemsPerRow, rem = item - rr_ * kItemsPerRow;
        const int c = rem / kFwdQuads, q = rem - c * kFwdQuads;
        const int row = row_first + rr_, col = c0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)row < (unsigned)H && (unsigned)col < (unsigned)W)
            v = *(const float4*)(X + c * ps + (size_t)row * W + col);
        const int slot = (slot_first + rr_) % kFwdRing;
        float* d = (float*)(ring + slot * kSlotBytes + c * kFwdRowBytes) + q;
        d[0] = v.x, d[kFwdQuads] = v.y, d[2 * kFwdQuads] = v.z, d[3 * kFwdQuads] = v.w;
    };
    // the tile's first window: several loads in flight per thread (one at a time, the start-up of a tile
    // is a chain of memory latencies, a sixth of the kernel's time)
    {
        constexpr int kTotal = kFwdSpan * kItemsPerRow, kBatch = 8;
        for (int first = tid; first < kTotal; first += kBatch * nthreads)
        {
            float4 v[kBatch];
            float* d[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u)
            {
                const int item = min(first + u * nthreads, kTotal - 1);
                const int rr_ = item / kItemsPerRow, rem = item - rr_ * kItemsPerRow;
                const int c = rem / kFwdQuads, q = rem - c * kFwdQuads;
                const int row = rbase + rr_, col = c0 + 4 * q;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((unsigned)row < (unsigned)H && (unsigned)col < (unsigned)W)
                    v[u] = *(const float4*)(X + c * ps + (size_t)row * W + col);
                d[u] = (float*)(ring + rr_ * kSlotBytes + c * kFwdRowBytes) + q;
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u)
                if (first + u * nthreads < kTotal)
                    d[u][0] = v[u].x, d[u][kFwdQuads] = v[u].y, d[u][2 * kFwdQuads] = v[u].z, d[u][3 * kFwdQuads] = v[u].w;
        }
    }

    // Pieces of this thread within a batch of four HR rows (fixed for the whole tile).  Blocks of
    // many waves (the common case) own at most kSlots pieces per batch and load them a whole LR row
    // before they are written to LDS; blocks of few waves stage piece by piece.
    // The waves of a block are dealt to the CU's four SIMDs in turn, so in a block of 4n+1 waves those
    // numbered 0, 4, 8 ... share one SIMD that carries a wave more than the others, and the block moves
    // at that SIMD's pace (one barrier per LR row): these waves take no part in the staging.
    constexpr int kSlots = 3;
    const int nwaves = nthreads >> 6, wv = tid >> 6;
    const bool skew = nwaves > 1 && (nwaves & 3) == 1;
    const bool stager = !skew || (wv & 3) != 0;
    const int n_st = skew ? nthreads - 64 * ((nwaves + 3) >> 2) : nthreads;
    const int stid = skew ? tid - 64 * ((wv >> 2) + 1) : tid;
    const bool early = kSlots * n_st >= 4 * kItemsPerRow;
    int p_rr[kSlots], p_lds[kSlots];
    size_t p_src[kSlots];
    bool p_own[kSlots], p_col[kSlots];
#pragma unroll
    for (int i = 0; i < kSlots; ++i)
    {
        const int item = max(stid, 0) + i * n_st;
        const int rr_ = item / kItemsPerRow, rem = item - rr_ * kItemsPerRow;
        const int c = rem / kFwdQuads, q = rem - c * kFwdQuads;
        const int col = c0 + 4 * q;
        p_own[i] = early && stager && item < 4 * kItemsPerRow;
        p_col[i] = (unsigned)col < (unsigned)W;
        p_rr[i] = rr_;
        p_lds[i] = c * kFwdRowBytes + 4 * q;
        p_src[i] = c * ps + (size_t)(p_col[i] ? col : 0);
    }
    // Through a buffer descriptor over all planes of X: a piece that is not this thread's, or lies outside the
    // image, is read at an offset beyond the buffer and comes back as zeros - no load sits under a branch.
    const __amdgpu_buffer_rsrc_t x_planes =
        __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((size_t)CH * ps * sizeof(float)), 0x00027000);
    auto batch_load = [&](int row_first, float4 (&v)[kSlots]) {
#pragma unroll
        for (int i = 0; i < kSlots; ++i)
        {
            const int row = row_first + p_rr[i];
            const bool real = p_own[i] && p_col[i] && (unsigned)row < (unsigned)H;
            const unsigned at = real ? (unsigned)((p_src[i] + (size_t)row * W) * sizeof(float)) : 0xFFFFFFF0u;
            const v4f q = __builtin_amdgcn_raw_buffer_load_b128(x_planes, (int)at, 0, 0);
            v[i] = make_float4(q.x, q.y, q.z, q.w);
        }
    };
    auto batch_put = [&](int slot_first, const float4 (&v)[kSlots]) {
#pragma unroll
        for (int i = 0; i < kSlots; ++i)
            if (p_own[i])
            {
                int slot = slot_first + p_rr[i];
                slot = (int)min((unsigned)slot, (unsigned)(slot - kFwdRing));
                float* d = (float*)(ring + slot * kSlotBytes + p_lds[i]);
                d[0] = v[i].x, d[kFwdQuads] = v[i].y, d[2 * kFwdQuads] = v[i].z, d[3 * kFwdQuads] = v[i].w;
            }
    };

    // The entries of LR row y: the long form for a row of kind mixed or outside, else the short form.
    const size_t grp_plane = (size_t)h * w;
    auto load_groups = [&](int y, uint8_t kind, uint2 (&m)[4]) {
        const bool lng = kind == kRowMixed || kind == kRowOutside;
        // Both words come through buffer descriptors whose size is zero when the word is not needed (no entry at
        // all for an identity row, no second word for a short row): such a load is answered with zeros without
        // touching memory.  (The same loads under branches make the compiler wait for every load in flight right
        // behind them, and a row that is followed by a long row then pays a whole memory latency.)
        const int plane_bytes = (int)(grp_plane * sizeof(uint4));
        const int at = (int)(((size_t)y * w + xl) * sizeof(uint4));
        const __amdgpu_buffer_rsrc_t first = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(grp_base + (lng ? grp_plane : 0)), 0, kind == kRowIdentity ? 0 : plane_bytes, 0x00027000);
        const __amdgpu_buffer_rsrc_t second = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(grp_base + 2 * grp_plane), 0, lng ? plane_bytes : 0, 0x00027000);
        const v4u b0 = __builtin_amdgcn_raw_buffer_load_b128(first, at, 0, 2 /* streaming */);
        const v4u b1 = __builtin_amdgcn_raw_buffer_load_b128(second, at, 0, 2);
        const uint4 a0 = make_uint4(b0.x, b0.y, b0.z, b0.w), a1 = make_uint4(b1.x, b1.y, b1.z, b1.w);
        m[0] = make_uint2(a0.x, a1.x), m[1] = make_uint2(a0.y, a1.y);
        m[2] = make_uint2(a0.z, a1.z), m[3] = make_uint2(a0.w, a1.w);
    };
    // The four pixels of one group (HR row N of this lane's LR column), every plane.  `dr0` is row
    // N's place in the current window, `s` the ring slot of the window's first row.  Returns false
    // when some pixel is not in the window (the values are then to be replaced by fetch_global).
    // The ring is the kernel's only LDS object, so its addresses are plain byte offsets.
    typedef const float __attribute__((address_space(3))) lds_cfloat;
    auto fetch = [&](auto form, const uint2 gm, int dr0, int s, bool checked, float4 (&a)[CH]) -> bool {
        typedef Grp<decltype(form)::value> G;
        const int dr = G::dr(gm) + dr0;
        const int dc = G::dc(gm) + dc_lane;
        // sources in other rows than the first pixel's are rare: one branch for the whole wave
        const bool rows_differ = __builtin_amdgcn_ballot_w64(G::rows_differ(gm)) != 0;
        unsigned far_row = (unsigned)dr, far_col = (unsigned)dc; // unsigned maxima: negatives count as huge
        float v[4][CH];
        auto pixel = [&](int j, unsigned row_at, unsigned e) {
            // element e of the row: sub-array e & 3, index e >> 2, i.e. byte e + (e & 3) * (4 * kFwdQuads - 1)
            unsigned at = __umul24(e & 3u, (unsigned)(4 * kFwdQuads - 1)) + (e + row_at);
            if (checked) // stay inside the ring, whatever the entry says
                at = min(at, (unsigned)(kRingBytes - (CH - 1) * kFwdRowBytes - 4));
#pragma unroll
            for (int c = 0; c < CH; ++c)
                v[j][c] = *(lds_cfloat*)(uintptr_t)(at + c * kFwdRowBytes);
        };
        // Ring rows as byte offsets: row (s + dr0 + d) of the ring starts at d * kSlotBytes + (s + dr0) * kSlotBytes,
        // less the ring's size when that runs past its end (one multiply-add, a subtraction and a minimum; the
        // second term is the same for the whole wave).
        const int ring_base = (s + dr0) * kSlotBytes;
        auto wrapped = [&](int at) { return min((unsigned)at, (unsigned)(at - kRingBytes)); };
        const int row0_at = __mul24(G::dr(gm), kSlotBytes) + ring_base; // not wrapped yet: the other rows start from it
        if (!rows_differ)
        {
            const unsigned row_at = wrapped(row0_at);
            pixel(0, row_at, (unsigned)dc);
            unsigned e1 = 0;
#pragma unroll
            for (int j = 1; j < 4; ++j)
            {
                unsigned e = (unsigned)(dc + j + G::dx(gm, j));
                if (checked)
                    far_col = max(far_col, e);
                if (j == 1)
                    e1 = e;
                if (j == 3)
                    e = mirror ? e1 : e;
                pixel(j, row_at, e);
            }
        }
        else
        {
            pixel(0, wrapped(row0_at), (unsigned)dc);
            unsigned e1 = 0, row1 = 0;
#pragma unroll
            for (int j = 1; j < 4; ++j)
            {
                const int dyj = G::dy(gm, j);
                unsigned e = (unsigned)(dc + j + G::dx(gm, j));
                if (checked)
                    far_row = max(far_row, (unsigned)(dr + dyj)), far_col = max(far_col, e);
                unsigned row_at = wrapped(__mul24(dyj, kSlotBytes) + row0_at);
                if (j == 1)
                    e1 = e, row1 = row_at;
                if (j == 3)
                    e = mirror ? e1 : e, row_at = mirror ? row1 : row_at;
                pixel(j, row_at, e);
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
            a[c] = make_float4(v[0][c], v[1][c], v[2][c], v[3][c]);
        return !checked ||
               (!G::far(gm) && far_row < (unsigned)kFwdSpan && far_col < (unsigned)(4 * kFwdQuads));
    };
    // (long entries)
    auto fetch_global = [&](const uint2 gm, int N, float4 (&a)[CH]) {
        typedef Grp<true> G;
        int col[4], row[4];
        if (G::far(gm))
        {
            const short2* m = (const short2*)raw_bwd.p[k] + (size_t)reflect101(N, H) * W + 4 * xl;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                col[j] = m[j].x, row[j] = m[j].y;
        }
        else
        {
            col[0] = 4 * xl + G::dc(gm);
            row[0] = N + G::dr(gm);
#pragma unroll
            for (int j = 1; j < 4; ++j)
            {
                col[j] = col[0] + j + G::dx(gm, j);
                row[j] = row[0] + G::dy(gm, j);
            }
        }
        float v[4][CH];
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const bool inside = (unsigned)col[j] < (unsigned)W && (unsigned)row[j] < (unsigned)H;
            const size_t idx = inside ? (size_t)row[j] * W + col[j] : (size_t)H * W; // the zero pixel
#pragma unroll
            for (int c = 0; c < CH; ++c)
                v[j][c] = X[c * ps + idx];
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
            a[c] = make_float4(v[0][c], v[1][c], v[2][c], mirror ? v[1][c] : v[3][c]);
    };
    // All four groups of an LR row from global memory: one unaligned 16-byte gather per group and plane
    // where the group is a run of pixels inside the image, pixel by pixel otherwise.
    auto fetch_global_rows = [&](const uint2 (&gm)[4], int N0, float4 (&a)[4][CH]) {
        typedef Grp<true> G;
        bool run[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const int col0 = 4 * xl + G::dc(gm[r]);
            const int row0 = N0 + r + G::dr(gm[r]);
            run[r] = G::run(gm[r]) && !G::far(gm[r]) && (unsigned)row0 < (unsigned)H && col0 >= 0 && col0 + 3 < W;
            const size_t idx = run[r] ? (size_t)row0 * W + col0 : (size_t)H * W;
#pragma unroll
            for (int c = 0; c < CH; ++c)
            {
                const float4_a4 v = *(const float4_a4*)(X + c * ps + idx);
                a[r][c] = make_float4(v.x, v.y, v.z, v.w);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (!run[r])
                fetch_global(gm[r], N0 + r, a[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c)
                a[r][c].w = mirror ? a[r][c].y : a[r][c].w;
    };
    auto row_pass = [&](const float4 (&a)[CH], float (&out)[CH]) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
        {
            const float l3 = from_left_lane(a[c].w), l2 = from_left_lane(a[c].z);
            float sum = a[c].x * k0;
            sum += (l3 + a[c].y) * k1;
            sum += (l2 + a[c].z) * k2;
            out[c] = sum;
        }
    };

    float rr[5][CH];
    const int ya = max(y0 - 1, 0);
    // long entry of the last group of the LR row above (checked below)
    const uint2 m0 = make_uint2(((const uint32_t*)(grp0 + grp_plane + (size_t)ya * w))[3],
                                ((const uint32_t*)(grp0 + 2 * grp_plane + (size_t)ya * w))[3]);
    uint2 mc[4];
    __syncthreads();
    {
        // HR row 4*y0 - 2 is the last of the LR row above (for y0 = 0 it reflects to row 2, the last of
        // LR row 0); its group's own row is N = 4*ya + 2
        const int N = 4 * ya + 2;
        float4 a[CH];
        if (!fetch(std::true_type{}, m0, N - rbase, 0, true, a))
            fetch_global(m0, N, a);
        row_pass(a, rr[0]);
    }

    // The rows that LR row y+1 adds to the window, its map entries and its LR pixel are loaded at
    // the top of row y and used at its end / in row y+1.  (Loading entries two rows ahead into alternating
    // register sets was tried and is slower: with loads under branches in the loop the compiler's wait counts
    // at the joins fall back to "all loads done", which puts the whole memory latency in front of every row.)
    int s = 0; // ring slot of the first row of the current LR row's window
    y_pack yv = *(const y_pack*)(yk + (size_t)y0 * w * CH);
    // The kinds of the tile's rows, 64 rows at a time, one per lane: a load per row would put a memory
    // latency in front of every row (the kind decides the row's code path).
    uint32_t kinds = flags[(size_t)min(y0 + lane, h - 1) * strips_x];
    uint8_t place = (uint8_t)(__builtin_amdgcn_readlane((int)kinds, 0) & kRowKindMask);
    load_groups(y0, place, mc);
    uint32_t sg_above = 0; // sign word of the LR row before the current one
#ifdef SRX_STAMPS
    unsigned long long t_compute = 0, t_put = 0, t_barrier = 0, t_prologue = SRX_TICK() - t_begin;
    unsigned n_place[4] = {0, 0, 0, 0};
#endif
    for (int y = y0; y < y1; ++y)
    {
#ifdef SRX_STAMPS
        const unsigned long long t_a = SRX_TICK();
        ++n_place[place & 3];
#endif
        const bool more = y + 1 < y1;
        const int ahead = y + 1 - y0; // row y + 1 within the tile
        if ((ahead & 63) == 0)
            kinds = flags[(size_t)min(y + 1 + lane, h - 1) * strips_x];
        const uint8_t place_next = (uint8_t)(__builtin_amdgcn_readlane((int)kinds, ahead & 63) & kRowKindMask);
        float4 sv[kSlots];
        batch_load(more ? 4 * y + 3 + kFwdMy : H, sv);
        uint2 mn[4];
        load_groups(min(y + 1, h - 1), place_next, mn);
        const y_pack yn = *(const y_pack*)(yk + (size_t)min(y + 1, h - 1) * w * CH);

        if (place == kRowIdentity)
        {
            // zero motion: pixel j of the group is element dc_lane + j of its own row, i.e. entry dc_lane / 4
            // of sub-array j - one address per row, everything else is in the instructions' offsets
            float4 a[4][CH];
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                int slot = s + kFwdMy + 1 + r;
                slot = slot >= kFwdRing ? slot - kFwdRing : slot;
                const unsigned at = (unsigned)(slot * kSlotBytes) + (unsigned)dc_lane;
#pragma unroll
                for (int c = 0; c < CH; ++c)
                {
                    a[r][c].x = *(lds_cfloat*)(uintptr_t)(at + c * kFwdRowBytes);
                    a[r][c].y = *(lds_cfloat*)(uintptr_t)(at + c * kFwdRowBytes + 4 * kFwdQuads);
                    a[r][c].z = *(lds_cfloat*)(uintptr_t)(at + c * kFwdRowBytes + 8 * kFwdQuads);
                    a[r][c].w = *(lds_cfloat*)(uintptr_t)(at + c * kFwdRowBytes + 12 * kFwdQuads);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    a[r][c].w = mirror ? a[r][c].y : a[r][c].w;
                row_pass(a[r], rr[r + 1]);
            }
        }
        else if (place == kRowOutside)
        {
            float4 a[4][CH];
            fetch_global_rows(mc, 4 * y - 1, a);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                row_pass(a[r], rr[r + 1]);
        }
        else if (place == kRowMixed)
        {
            // the lanes that find their pixels in the window read them there; the others (a few) gather from
            // global memory, all four groups' loads in flight before the first row pass
            float4 a[4][CH];
            bool found[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                found[r] = fetch(std::true_type{}, mc[r], kFwdMy + 1 + r, s, true, a[r]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (!found[r])
                    fetch_global(mc[r], 4 * y - 1 + r, a[r]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                row_pass(a[r], rr[r + 1]);
        }
        else
        {
            // all LDS reads of the LR row in flight before the first row pass
            float4 a[4][CH];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                fetch(std::false_type{}, mc[r], kFwdMy + 1 + r, s, false, a[r]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                row_pass(a[r], rr[r + 1]);
        }
        place = place_next;

        uint32_t sg = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c)
        {
            float sum = rr[2][c] * k0;
            sum += (rr[1][c] + rr[3][c]) * k1;
            sum += (rr[0][c] + rr[4][c]) * k2;
            sg |= (uint32_t)((int)dsign((float)yv.v[c], sum) + 1) << (10 * c);
        }
        // Cell (y-1) = {S(y-1), S(y)} leaves as one 8-byte store once both rows are known (whole cache
        // lines per wave); only the cells at the tile's first and last row are written in halves, the other
        // half coming from the tile above / below.
        if (owner)
        {
            if (y > y0)
                stream_store((uint2*)(sp + 2 * (size_t)(y - 1) * w), sg_above, sg);
            else if (y > 0)
                stream_store(sp + 2 * (size_t)(y - 1) * w + 1, sg);
            if (!more)
                stream_store(sp + 2 * (size_t)y * w, sg);
        }
        sg_above = sg;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            rr[0][c] = rr[4][c];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            mc[r] = mn[r];
        yv = yn;
#ifdef SRX_STAMPS
        const unsigned long long t_b = SRX_TICK();
#endif
        if (more)
        {
            // rows 4y+3+My .. 4y+6+My join the window; they start kFwdSpan rows below its first row
            int slot_first = s + kFwdSpan;
            slot_first = slot_first >= kFwdRing ? slot_first - kFwdRing : slot_first;
            batch_put(slot_first, sv);
            if (!early)
                for (int item = tid; item < 4 * kItemsPerRow; item += nthreads)
                    stage_piece(item, 4 * y + 3 + kFwdMy, slot_first);
        }
        s = s + 4 >= kFwdRing ? s + 4 - kFwdRing : s + 4;
#ifdef SRX_STAMPS
        const unsigned long long t_c = SRX_TICK();
#endif
        __syncthreads();
#ifdef SRX_STAMPS
        t_compute += t_b - t_a, t_put += t_c - t_b, t_barrier += SRX_TICK() - t_c;
#endif
    }
#ifdef SRX_STAMPS
    SRX_FWD_STAMPS(t_prologue, t_compute, t_put, t_barrier, SRX_TICK() - t_begin);
    if ((threadIdx.x & 63) == 0)
    {
        const unsigned wave_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (wave_ < (1u << 16)) // rows by kind, 16 bits each: mixed, inside, outside, identity
            g_stamps[wave_ * kStampSlots + 7] = n_place[0] | ((unsigned long long)n_place[1] << 16) |
                                                ((unsigned long long)n_place[2] << 32) | ((unsigned long long)n_place[3] << 48);
    }
#endif
}

// value held by lane+1 of the wave (0 in lane 63)
__device__ __forceinline__ float from_right_lane(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, true));
}

// Forward chain at scale 2, 5-tap blur, planar X: one wave per (strip of 62 LR columns, group of LR
// rows, window frame).  Lane l works on LR column x = x0 - 1 + l and holds, per HR row and plane, the
// warped pixels of HR columns 2x, 2x+1 (one 8-byte gather where the two sources are neighbours); the
// row pass centred on column 2x is
//     a0*k0 + (L.a1 + a1)*k1 + (L.a0 + R.a0)*k2
// with L, R the lanes to the left and right (lanes 0 and 63 only serve as neighbours).  reflect101 at
// the image edges: the left neighbour of column 0 delivers columns 2 and 1 in the places of -2 and
// -1; the right neighbour of the last column is a clamped copy of it, whose first pixel is column
// W-2, where W reflects to.  Rows are reflected when they are fetched.  The wave walks down its LR
// rows keeping the five row-pass results of the current LR row in registers (two new HR rows per LR
// row).  Writes plain sign planes.
#ifndef SRX_A2S_ROWS
#define SRX_A2S_ROWS 12
#endif
constexpr int kFwd2Strip = 62, kFwd2Rows = SRX_A2S_ROWS;

template <int CH, bool Y8>
__global__ __launch_bounds__(64) void k_forward_sign_s2(const float* __restrict__ X, int H, int W, size_t ps,
                                                        PtrList bwd_maps, PtrList ys, MutPtrList signs, int K, GaussK g,
                                                        int strips_x, unsigned n_items)
{
    const unsigned item = xcd_contiguous(blockIdx.x, n_items);
    if (item >= n_items)
        return;
    const int k = item % K;
    const unsigned tile = item / K;
    const int strip = tile % strips_x, rg = tile / strips_x;
    const int h = H >> 1, w = W >> 1;
    const int lane = threadIdx.x;
    const int x0 = strip * kFwd2Strip, y0 = rg * kFwd2Rows;
    const int y1 = min(y0 + kFwd2Rows, h);
    const int x = x0 - 1 + lane;
    const int xl = min(max(x, 0), w - 1);
    const bool owner = lane >= 1 && lane <= kFwd2Strip && x < w;
    const bool mirror = x == -1; // delivers columns 2 and 1
    const int col0 = mirror ? 2 : 2 * xl, col1 = mirror ? 1 : 2 * xl + 1;
    const short2* map = (const short2*)bwd_maps.p[k];
    typedef typename std::conditional<Y8, uint8_t, float>::type y_elem; // Y8: the 8-bit originals of the LR frames
    const y_elem* yk = (const y_elem*)ys.p[k] + (size_t)xl * CH;
    uint32_t* sp = (uint32_t*)signs.p[k] + xl;
    const float k0 = g.k[2], k1 = g.k[3], k2 = g.k[4];
    const unsigned zero_px = (unsigned)(H * W);

    // row pass of HR row Y (reflected into the image)
    auto hr_row = [&](int Y, float (&out)[CH]) {
        const short2* mrow = map + (size_t)reflect101(Y, H) * W;
        const short2 m0 = mrow[col0], m1 = mrow[col1];
        const bool in0 = (unsigned)(int)m0.x < (unsigned)W && (unsigned)(int)m0.y < (unsigned)H;
        const bool in1 = (unsigned)(int)m1.x < (unsigned)W && (unsigned)(int)m1.y < (unsigned)H;
        const unsigned i0 = in0 ? (unsigned)((int)m0.y * W + (int)m0.x) : zero_px;
        const unsigned i1 = in1 ? (unsigned)((int)m1.y * W + (int)m1.x) : zero_px;
        float a0[CH], a1[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c)
        {
            const packed_floats<2> v = *(const packed_floats<2>*)(X + c * ps + i0); // the zero pixel is padded
            a0[c] = v.v[0], a1[c] = v.v[1];
        }
        if (i1 != i0 + 1 || !in0)
        {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                a1[c] = X[c * ps + i1];
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
        {
            const float l1 = from_left_lane(a1[c]), l0 = from_left_lane(a0[c]), r0 = from_right_lane(a0[c]);
            float sum = a0[c] * k0;
            sum += (l1 + a1[c]) * k1;
            sum += (l0 + r0) * k2;
            out[c] = sum;
        }
    };

    float rr[5][CH];
    hr_row(2 * y0 - 2, rr[0]);
    hr_row(2 * y0 - 1, rr[1]);
    hr_row(2 * y0, rr[2]);
    for (int y = y0; y < y1; ++y)
    {
        float yv[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c)
            yv[c] = (float)yk[(size_t)y * w * CH + c];
        hr_row(2 * y + 1, rr[3]);
        hr_row(2 * y + 2, rr[4]);
        uint32_t sg = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c)
        {
            float sum = rr[2][c] * k0;
            sum += (rr[1][c] + rr[3][c]) * k1;
            sum += (rr[0][c] + rr[4][c]) * k2;
            sg |= (uint32_t)((int)dsign(yv[c], sum) + 1) << (10 * c);
        }
        if (owner)
            stream_store(sp + (size_t)y * w, sg);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            rr[0][c] = rr[2][c], rr[1][c] = rr[3][c], rr[2][c] = rr[4][c];
    }
}

// The same chain with two LR columns (four HR columns) per lane, for images whose HR width is a multiple
// of 4.  A vector load costs the CU's address unit 16 cycles per wave whatever its width (only a wave of
// consecutive dwords is cheaper), and the kernel above is bound by exactly that; this one reads a lane's
// four map entries of an HR row with one 16-byte load and their pixels with one 16-byte load per plane
// where the four sources are consecutive in X.  Where they are not (a rounding step of the motion inside
// the lane's four pixels), a second 16-byte load ending at the fourth pixel's source covers every pattern
// with one step; other patterns load pixel by pixel.  Pixels outside the image read the four zero pixels
// behind the plane.  Lane l holds HR columns c0 .. c0+3, c0 = 2*x0 - 4 + 4*l; lanes 0 and 63 only serve as
// neighbours: lane 0 of the first strip reads columns 0..3 and delivers columns 2, 1 in the places of -2, -1,
// the lane at c0 = W reads columns W-4 .. W-1 and delivers column W-2 in the place of W.  The map entries and
// the LR pixels of the next LR row are loaded before the pixels of the current one are gathered.
constexpr int kFwd2Cols = 124;
// SRX_FWD_S2_NARROW in the environment selects the one-column-per-lane kernel (for comparisons)
static const bool g_forward_s2_wide = std::getenv("SRX_FWD_S2_NARROW") == nullptr;

template <int CH, bool Y8> struct YPair;
template <> struct YPair<1, true>
{
    uint16_t v;
    __device__ __forceinline__ float get(int j, int) const { return (float)((v >> (8 * j)) & 0xFF); }
};
template <> struct __attribute__((packed, aligned(2))) YPair<3, true>
{
    uint32_t lo;
    uint16_t hi;
    __device__ __forceinline__ float get(int j, int c) const
    {
        const int at = 3 * j + c;
        return (float)(at < 4 ? (lo >> (8 * at)) & 0xFF : (hi >> (8 * (at - 4))) & 0xFF);
    }
};
template <int CH> struct __attribute__((packed, aligned(4))) YPair<CH, false>
{
    float v[2 * CH];
    __device__ __forceinline__ float get(int j, int c) const { return v[CH * j + c]; }
};

#ifndef SRX_S2W_WAVES
#define SRX_S2W_WAVES 4
#endif
template <int CH, bool Y8>
__global__ __launch_bounds__(64, SRX_S2W_WAVES) void k_forward_sign_s2w(const float* __restrict__ X, int H, int W, size_t ps,
                                                         PtrList bwd_maps, PtrList ys, MutPtrList signs, int K, GaussK g,
                                                         int strips_x, unsigned n_items)
{
    const unsigned item = xcd_contiguous(blockIdx.x, n_items);
    if (item >= n_items)
        return;
    const int k = item % K;
    const unsigned tile = item / K;
    const int strip = tile % strips_x, rg = tile / strips_x;
    const int h = H >> 1, w = W >> 1;
    const int lane = threadIdx.x;
    const int x0 = strip * kFwd2Cols, y0 = rg * kFwd2Rows;
    const int y1 = min(y0 + kFwd2Rows, h);
    const int xa = x0 - 2 + 2 * lane; // the lane's LR columns are xa, xa + 1
    const int c0 = 2 * xa;
    const bool left = c0 < 0, right = c0 == W;
    const int cl = min(max(c0, 0), W - 4);
    const int xl = min(max(xa, 0), w - 2);
    const bool owner = lane >= 1 && lane <= kFwd2Cols / 2 && xa < w;
    const uint4* map = (const uint4*)((const short2*)bwd_maps.p[k] + cl);
    typedef typename std::conditional<Y8, uint8_t, float>::type y_elem;
    typedef YPair<CH, Y8> y_pair;
    const y_elem* yk = (const y_elem*)ys.p[k] + (size_t)xl * CH;
    uint32_t* sp = (uint32_t*)signs.p[k] + xl;
    const float k0 = g.k[2], k1 = g.k[3], k2 = g.k[4];
    const unsigned zero_px = (unsigned)(H * W);
    const int row_quads = W >> 2;

    auto entries = [&](int Y) { return stream_load(map + (size_t)reflect101(Y, H) * row_quads); };
    // An HR row in two steps, so that the gathers of the two HR rows of an LR row are all in flight before the
    // first of them is waited for: gathers from the lane's four map entries, then the row pass: out[0] for LR
    // column xa, out[1] for xa + 1.
    struct Row
    {
        unsigned i[4];
        bool ok1, ok2, ok3, b1, b2, b3;
        packed_floats<4> v[CH], t[CH];
    };
    auto hr_issue = [&](const uint4 e, Row& r) {
        const uint32_t ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const int mx = (int)(short)(ev[j] & 0xFFFF), my = (int)ev[j] >> 16;
            const bool in = (unsigned)mx < (unsigned)W && (unsigned)my < (unsigned)H;
            r.i[j] = in ? (unsigned)(my * W + mx) : zero_px + j;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
            r.v[c] = *(const packed_floats<4>*)(X + c * ps + r.i[0]);
        r.ok1 = r.i[1] == r.i[0] + 1, r.ok2 = r.i[2] == r.i[0] + 2, r.ok3 = r.i[3] == r.i[0] + 3;
        // the second gather, which ends at the fourth source, by the lanes whose sources are not one run
        const unsigned ib = max(r.i[3], 3u) - 3;
        r.b1 = r.i[1] == ib + 1, r.b2 = r.i[2] == ib + 2, r.b3 = r.i[3] == ib + 3;
        if (!(r.ok1 && r.ok2 && r.ok3))
        {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                r.t[c] = *(const packed_floats<4>*)(X + c * ps + ib);
        }
    };
    auto hr_finish = [&](Row& r, float (&out)[2][CH]) {
        float a[4][CH];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                a[j][c] = r.v[c].v[j];
        if (!(r.ok1 && r.ok2 && r.ok3))
        {
#pragma unroll
            for (int c = 0; c < CH; ++c)
            {
                a[1][c] = r.ok1 ? a[1][c] : r.t[c].v[1];
                a[2][c] = r.ok2 ? a[2][c] : r.t[c].v[2];
                a[3][c] = r.ok3 ? a[3][c] : r.t[c].v[3];
            }
        }
        if (__builtin_amdgcn_ballot_w64(!((r.ok1 || r.b1) && (r.ok2 || r.b2) && (r.ok3 || r.b3))) != 0)
        {
            if (!(r.ok1 || r.b1))
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    a[1][c] = X[c * ps + r.i[1]];
            if (!(r.ok2 || r.b2))
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    a[2][c] = X[c * ps + r.i[2]];
            if (!(r.ok3 || r.b3))
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    a[3][c] = X[c * ps + r.i[3]];
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
        {
            a[3][c] = left ? a[1][c] : a[3][c];
            a[0][c] = right ? a[2][c] : a[0][c];
            const float l3 = from_left_lane(a[3][c]), l2 = from_left_lane(a[2][c]), r0 = from_right_lane(a[0][c]);
            float sum = a[0][c] * k0;
            sum += (l3 + a[1][c]) * k1;
            sum += (l2 + a[2][c]) * k2;
            out[0][c] = sum;
            sum = a[2][c] * k0;
            sum += (a[1][c] + a[3][c]) * k1;
            sum += (a[0][c] + r0) * k2;
            out[1][c] = sum;
        }
    };
    auto hr_row = [&](const uint4 e, float (&out)[2][CH]) {
        Row r;
        hr_issue(e, r);
        hr_finish(r, out);
    };

    float rr[5][2][CH];
    hr_row(entries(2 * y0 - 2), rr[0]);
    hr_row(entries(2 * y0 - 1), rr[1]);
    hr_row(entries(2 * y0), rr[2]);
    uint4 e3 = entries(2 * y0 + 1), e4 = entries(2 * y0 + 2);
    y_pair yv = *(const y_pair*)(yk + (size_t)y0 * w * CH);
    for (int y = y0; y < y1; ++y)
    {
        const int yn = min(y + 1, h - 1);
        const uint4 n3 = entries(2 * yn + 1), n4 = entries(2 * yn + 2);
        const y_pair ynext = *(const y_pair*)(yk + (size_t)yn * w * CH);
        {
            Row ra, rb;
            hr_issue(e3, ra);
            hr_issue(e4, rb);
            hr_finish(ra, rr[3]);
            hr_finish(rb, rr[4]);
        }
        uint32_t sg[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < CH; ++c)
            {
                float sum = rr[2][j][c] * k0;
                sum += (rr[1][j][c] + rr[3][j][c]) * k1;
                sum += (rr[0][j][c] + rr[4][j][c]) * k2;
                sg[j] |= (uint32_t)((int)dsign(yv.get(j, c), sum) + 1) << (10 * c);
            }
        if (owner)
            stream_store((uint2*)(sp + (size_t)y * w), sg[0], sg[1]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < CH; ++c)
                rr[0][j][c] = rr[2][j][c], rr[1][j][c] = rr[3][j][c], rr[2][j][c] = rr[4][j][c];
        e3 = n3, e4 = n4, yv = ynext;
    }
}

// Integer weights of the transposed 5-tap blur for dyadic kernels (k*16 integer):
// per weight-table index the weights of the 2x2 block of sign cells, in units of
// 1/256 and in the order of a pair-plane read, and their sum (the last entry is
// all zero).  `flush` = number of frames whose weighted sign words can be summed in
// the packed 10-bit fields before they must be unpacked.
struct FastW
{
    // scale 4: [0..3] = cells (row, col), (row+1, col), (row, col+1), (row+1, col+1) of a pair-plane read;
    // scale 2: [3*dy + dx] = cell (row+dy, col+dx) of a plain plane, dy, dx in 0..2
    uint32_t w[kFwdLutSize][9];
    uint32_t wsum[kFwdLutSize];
    int flush;
};
// 12 bytes at any 4-byte aligned address: one global_load_dwordx3
struct __attribute__((packed, aligned(4))) uint3_a4
{
    uint32_t x, y, z;
};

// 16 bytes at an 8-byte aligned address: one global_load_dwordx4
struct __attribute__((packed, aligned(8))) uint4_a8
{
    uint32_t x, y, z, w;
};

// Comparisons of pixel pairs for the BTV term, two pixels per instruction.  The BTV tile holds the pixels
// times 2^64 (exact for |pixel| < 2^63).  A difference d = a - b of two such values is +-0 exactly when
// a == b, has the sign of the comparison otherwise, and is never smaller in magnitude than 2^-85 (the
// smallest gap between two floats, 2^-149, times 2^64).  d * 2^86 + 0.5, clamped to [0, 1] by the
// instruction's output modifier, is therefore 1, 0.5 or 0 for a > b, a == b, a < b: (sign(a - b) + 1) / 2
// from one packed multiply-add.  The reference's sign(a - b) - sign(c - a) (lib/src/btv_l1.cpp:221-233) is
// then 2 * (half_step(a - b) - half_step(c - a)); the BTV weights are passed times 2 (exact), so that the
// multiply-add of the accumulation forms the reference's product, still exact, and rounds as before.
// A tile whose scaled values are not all finite (NaN or infinite pixels, magnitudes of 2^63 and more)
// takes the comparison path on the unscaled values instead, which is the reference's diffSign
// (lib/src/btv_l1.cpp:138-141) literally.
constexpr float kBtvTileScale = 0x1p64f, kBtvStepScale = 0x1p86f;
__device__ __forceinline__ float btv_tile_value(float x)
{
    return x * kBtvTileScale;
}
// (sign(d) + 1) / 2 for both halves of d; `scale2` holds kBtvStepScale twice
__device__ __forceinline__ v2f half_step2(v2f d, v2f scale2)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(r) : "v"(d), "v"(scale2));
    return r;
}
__device__ __forceinline__ float half_step(float d, float scale)
{
    float r;
    asm("v_fma_f32 %0, %1, %2, 0.5 clamp" : "=v"(r) : "v"(d), "v"(scale));
    return r;
}
// a * b + c on the low 24 bits of a and b: one v_mad_u32_u24 (the compiler prefers two multiplies and a
// three-way add for a pair of these)
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Back-projection of all K frames + BTV regulariser + update for one 64x16 tile
// of HR pixels per block; each thread owns 4 consecutive pixels of one row.
// X and X_out are planar.
//  * BTV, one channel after the other: the channel's tile of X (+3 px halo) is
//    staged in LDS (the next channel's loads are in flight meanwhile); the
//    comparisons come from half_step2 above, two pixels per instruction, or
//    from plain compares for a tile whose scaled values are not all finite;
//    accumulation order is the reference's (lib/src/btv_l1.cpp:221-233).  The
//    kernel is bound by the issue of vector instructions (one per SIMD every
//    four cycles, whatever the instruction), so what counts is their number.
//    The results rest in LDS / registers while the frame loop runs.
//  * per frame: one 16-byte load of 4 decoded forward-map entries (fetched one
//    frame ahead), then per pixel one 16-byte read of the 2x2 block of sign
//    cells from the pair plane, times the entry's integer weights.  A sign word
//    times a weight is three products at once (10-bit fields), and the products
//    of several frames are summed before the fields are unpacked.  Exact for
//    dyadic blur kernels: every partial sum is a multiple of 1/256.  The loop is
//    branch-free: entries outside the image carry zero weights, and the
//    targets next to the image edge have weight-table entries of their own.
//  * scale 2 (FOOT == 3): a target touches 3 x 3 cells of a plain sign plane.  The patch of every frame's
//    plane that the tile's pixels reach is staged in LDS (kSgRows x kSgCols cells, placed per tile and frame by
//    k_tile_relative, which also makes the map entries relative to it), one frame ahead of its use, one barrier
//    per frame; the nine cells per pixel are then LDS reads.  Gathered from global memory they took 27 cycles
//    of the texture path per 64-lane instruction and the kernel ran at that path's pace.  A tile whose entries
//    do not fit the patch (header 0) gathers from global memory as before.
//  * lanes are mapped to tile positions so that each lane group of a
//    ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) reads 16
//    consecutive 16-byte chunks of one LDS row: conflict-free at any row pitch.
#ifndef SRX_B_WAVES
#define SRX_B_WAVES 6
#endif
// waves per SIMD of the scale-2 variant (it holds two sets of staged sign cells besides the map entries)
#ifndef SRX_B3_WAVES
#define SRX_B3_WAVES 5
#endif
// the thread's own pixels read again for the update (1) instead of kept in registers across the frame loop (0)
#ifndef SRX_B3_REREAD
#define SRX_B3_REREAD 1
#endif
// frames ahead that the staged sign cells are asked for (scale-2 variant): 1 or 2
#ifndef SRX_B3_DEPTH
#define SRX_B3_DEPTH 1
#endif
// pixels whose LDS reads of sign cells are in flight together (scale-2 variant)
#ifndef SRX_B3_LDS_PX
#define SRX_B3_LDS_PX 2
#endif
#ifndef SRX_B_PARK
#define SRX_B_PARK 2
#endif
// threads per block; a block covers 64 x (threads / 16) pixels
#ifndef SRX_B_THREADS
#define SRX_B_THREADS 256
#endif

// FOOT = cells per axis a target can touch: 2 (scale 4, pair plane) or 3 (scale 2, plain plane)
template <int CH, int FOOT>
__global__ __launch_bounds__(SRX_B_THREADS, FOOT == 2 ? SRX_B_WAVES : SRX_B3_WAVES) void k_backproject_update_s4(
    const float* __restrict__ X, float* __restrict__ X_out, int H, int W, size_t ps, PtrList dec_fwd,
    PtrList signs, int K, FastW fw, BtvW bw, double tau, double lambda, int tiles_x, unsigned n_tiles,
    const void* __restrict__ still_signs)
{
    constexpr int kThreads = SRX_B_THREADS, TYN = kThreads / 16, PXT = 4;
    constexpr int TW = 16 * PXT;        // 64 px
    constexpr int HALO = 3, HL = 4;     // halo; left halo padded to 4 px so that row loads are 16-byte aligned
    constexpr int PITCH = TW + HL + 4;  // 72 floats per LDS row: columns [j0-4, j0+68)
    constexpr int LROWS = TYN + 2 * HALO;
    constexpr int RV = PITCH / 4;       // 16-byte vectors per LDS row
    constexpr int NV = 3;               // vectors per window row: columns [4tx, 4tx+12) of the LDS row
    __shared__ __attribute__((aligned(16))) float tile[LROWS * PITCH]; // one channel at a time
    // scale 4: [li] = weights, [kFwdLutSize + li].x = their sum; scale 2: [3*li .. 3*li+2] = nine weights, sum, -, -
    __shared__ uint4 lut[(FOOT == 2 ? 2 : 3) * kFwdLutSize];
    // regulariser values rest while the frame loop runs: the first kPark channels in LDS, the others in registers
    constexpr int kPark = SRX_B_PARK < CH ? SRX_B_PARK : CH;
    __shared__ float park[(kPark > 0 ? kPark : 1) * PXT * kThreads];
    float keep[CH][PXT];
    float centre[CH][PXT];

    const unsigned t = xcd_contiguous(blockIdx.x, n_tiles);
    if (t >= n_tiles)
        return;
    const int j0 = (t % tiles_x) * TW, i0 = (t / tiles_x) * TYN;
    const int tid = threadIdx.x;
    // lane -> (tx, row within the wave's 4 rows), see the note on ds_read_b128 lane groups above
    int tx, ty;
    {
        const int l5 = tid & 31;
        const int grp = (0x96 >> (l5 >> 2)) & 1; // lanes 0-3,12-15,20-27 -> 0; 4-11,16-19,28-31 -> 1
        const int pos = l5 < 4 ? l5 : l5 < 12 ? l5 - 4 : l5 < 20 ? l5 - 8 : l5 < 28 ? l5 - 12 : l5 - 16;
        tx = pos;
        ty = (tid >> 6) * 4 + ((tid >> 5) & 1) * 2 + grp;
    }
    const int i = min(i0 + ty, H - 1), j = min(j0 + tx * PXT, W - PXT); // clamped; stores are masked below
    const bool live = (i0 + ty) < H && (j0 + tx * PXT) < W; // W is a multiple of 4: a thread's pixels are all in or all out
    const size_t pix = (size_t)i * W + j;

    SRX_STAMP(0);
    // first frame's map entries: issued before the tile is staged
    uint4 cur = make_uint4(0, 0, 0, 0);
    if (K > 0)
        cur = stream_load((const uint4*)((const uint32_t*)dec_fwd.p[0] + pix));
    // scale 2: the 16-bit entries of a tile that has them (k_tile_relative); the first frame's are fetched in both
    // forms, the tile's header is not known yet
    const size_t short_entries = (size_t)H * W + (((size_t)n_tiles + 1) & ~(size_t)1); // in words, from the map's start
    uint2 cur16 = make_uint2(0, 0);
    if (FOOT == 3 && K > 0)
        cur16 = stream_load((const uint2*)((const uint32_t*)dec_fwd.p[0] + short_entries) + pix / 4);

    if (tid < kFwdLutSize)
    {
        if (FOOT == 2)
        {
            lut[tid] = make_uint4(fw.w[tid][0], fw.w[tid][1], fw.w[tid][2], fw.w[tid][3]);
            lut[kFwdLutSize + tid] = make_uint4(fw.wsum[tid], 0, 0, 0);
        }
        else
        {
            lut[3 * tid] = make_uint4(fw.w[tid][0], fw.w[tid][1], fw.w[tid][2], fw.w[tid][3]);
            lut[3 * tid + 1] = make_uint4(fw.w[tid][4], fw.w[tid][5], fw.w[tid][6], fw.w[tid][7]);
            lut[3 * tid + 2] = make_uint4(fw.w[tid][8], fw.wsum[tid], 0, 0);
        }
    }
    // ---- BTV regulariser, one channel at a time: the channel's tile of X (+3 px halo) is staged in LDS
    // (branch-free clamped loads, the next channel's issued before this channel's arithmetic starts)
    constexpr int NCH1 = LROWS * RV; // 16-byte chunks of one channel's tile
    constexpr int NIT1 = (NCH1 + kThreads - 1) / kThreads;
    float4 stage[NIT1];
    auto load_stage = [&](int c) {
#pragma unroll
        for (int it = 0; it < NIT1; ++it)
        {
            const int idx = min(tid + it * kThreads, NCH1 - 1);
            const int r = idx / RV, c4 = idx - r * RV;
            const int gi = min(max(i0 - HALO + r, 0), H - 1), gj = min(max(j0 - HL + 4 * c4, 0), W - 4);
            stage[it] = *(const float4*)(X + c * ps + (size_t)gi * W + gj);
        }
    };
    const int lrow = ty + HALO;
    // one verdict per wave and channel: is every scaled value of the wave's share of the tile finite
    __shared__ uint32_t odd_values[CH][kThreads / 64];
    __shared__ float whole_weights[22]; // for the tiles that take the comparison path
    typedef const float __attribute__((address_space(3))) lds_cfloat;
    if (tid >= 64 && tid < 64 + 22)
        whole_weights[tid - 64] = bw.w[tid - 64] * 0.5f;
    if (lambda > 0)
    {
        load_stage(0);
        const v2f step2 = v2f{kBtvStepScale, kBtvStepScale};
#pragma unroll
        for (int c = 0; c < CH; ++c)
        {
            // 0 * value summed over the thread's share: 0 while every scaled value is finite, NaN otherwise
            v2f weird = v2f{0.0f, 0.0f};
#pragma unroll
            for (int it = 0; it < NIT1; ++it)
            {
                const int idx = tid + it * kThreads;
                if (idx < NCH1)
                {
                    const float4 v = stage[it];
                    const v2f lo = v2f{btv_tile_value(v.x), btv_tile_value(v.y)}, hi = v2f{btv_tile_value(v.z), btv_tile_value(v.w)};
                    weird = __builtin_elementwise_fma(lo, v2f{0.0f, 0.0f}, weird);
                    weird = __builtin_elementwise_fma(hi, v2f{0.0f, 0.0f}, weird);
                    // cells outside the image hold copies of edge pixels (the loads are clamped); only pixels
                    // within HALO of the image edge read them, and their regulariser is discarded below
                    *(float4*)&tile[4 * idx] = make_float4(lo.x, lo.y, hi.x, hi.y);
                }
            }
            {
                const bool odd = !(weird.x == 0.0f) || !(weird.y == 0.0f);
                const bool any = __builtin_amdgcn_ballot_w64(odd) != 0;
                if ((tid & 63) == 0)
                    odd_values[c][tid >> 6] = any ? 1u : 0u;
            }
            __syncthreads();
            if (c == 0)
                SRX_STAMP(1);
            bool plain = true;
#pragma unroll
            for (int wv = 0; wv < kThreads / 64; ++wv)
                plain = plain && odd_values[c][wv] == 0u;
            v2f reg01 = v2f{0.0f, 0.0f}, reg23 = v2f{0.0f, 0.0f};
            if (__builtin_expect(plain, 1))
            {
                if (c + 1 < CH)
                    load_stage(c + 1);
                lds_cv4i* tp = (lds_cv4i*)tile + lrow * RV + tx; // window start of the centre row
                v4i cv[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    cv[v] = tp[v];
                const float* cf = (const float*)cv;
                // the thread's own pixels, unscaled again (exact), for the update at the end
#pragma unroll
                for (int e = 0; e < PXT; ++e)
                    centre[c][e] = cf[HL + e] * (1.0f / kBtvTileScale);
                const v2f c01 = *(const v2f*)&cf[HL], c23 = *(const v2f*)&cf[HL + 2], c12 = v2f{cf[HL + 1], cf[HL + 2]};
                int ind = 0;
#pragma unroll
                for (int m = 0; m <= 3; ++m)
                {
                    v4i pv[NV], mv[NV];
#pragma unroll
                    for (int v = 0; v < NV; ++v)
                    {
                        pv[v] = m == 0 ? cv[v] : tp[m * RV + v];
                        mv[v] = m == 0 ? cv[v] : tp[-m * RV + v];
                    }
                    const float* pf = (const float*)pv;
                    const float* mf = (const float*)mv;
#pragma unroll
                    for (int l = 3; l + m >= 0; --l, ++ind)
                    {
                        if (m == 0 && l == 0)
                            continue; // both signs compare the pixel with itself: adds w * 0
                        const float wt = bw.w[ind]; // twice the weight
                        // fp[e] = half_step(X(p) - X(p + d)), bp[e] = half_step(X(p - d) - X(p)) for the
                        // thread's pixels e = 0..3.  The window registers pair up for pixels (0,1), (2,3) when
                        // l is even - subtraction and step are then packed - and for pixels (1,2) when l is odd:
                        // packed subtraction, and scalar steps that write into the pairs (0,1), (2,3).
                        float fp[PXT], bp[PXT];
                        if ((l & 1) == 0)
                        {
                            const v2f a = half_step2(c01 - *(const v2f*)&pf[HL + l], step2);
                            const v2f b = half_step2(c23 - *(const v2f*)&pf[HL + 2 + l], step2);
                            fp[0] = a.x, fp[1] = a.y, fp[2] = b.x, fp[3] = b.y;
                        }
                        else
                        {
                            const v2f a = c12 - *(const v2f*)&pf[HL + 1 + l];
                            fp[1] = half_step(a.x, kBtvStepScale), fp[2] = half_step(a.y, kBtvStepScale);
                            fp[0] = half_step(cf[HL] - pf[HL + l], kBtvStepScale);
                            fp[3] = half_step(cf[HL + 3] - pf[HL + 3 + l], kBtvStepScale);
                        }
                        if (m == 0)
                        {
                            // within the row X(p - d) - X(p) is the forward difference of pixel e - l, which
                            // this thread has when e >= l
#pragma unroll
                            for (int e = 0; e < PXT; ++e)
                                bp[e] = e >= l ? fp[e - l] : half_step(mf[HL + e - l] - cf[HL + e], kBtvStepScale);
                        }
                        else if ((l & 1) == 0)
                        {
                            const v2f a = half_step2(*(const v2f*)&mf[HL - l] - c01, step2);
                            const v2f b = half_step2(*(const v2f*)&mf[HL + 2 - l] - c23, step2);
                            bp[0] = a.x, bp[1] = a.y, bp[2] = b.x, bp[3] = b.y;
                        }
                        else
                        {
                            const v2f a = *(const v2f*)&mf[HL + 1 - l] - c12;
                            bp[1] = half_step(a.x, kBtvStepScale), bp[2] = half_step(a.y, kBtvStepScale);
                            bp[0] = half_step(mf[HL - l] - cf[HL], kBtvStepScale);
                            bp[3] = half_step(mf[HL + 3 - l] - cf[HL + 3], kBtvStepScale);
                        }
                        // (2 w) * (fp - bp) = w * (sign - sign) exactly, so the fused form rounds once, like
                        // the reference's acc += w * d
                        reg01 = __builtin_elementwise_fma(v2f{wt, wt}, v2f{fp[0], fp[1]} - v2f{bp[0], bp[1]}, reg01);
                        reg23 = __builtin_elementwise_fma(v2f{wt, wt}, v2f{fp[2], fp[3]} - v2f{bp[2], bp[3]}, reg23);
                    }
                }
            }
            else
            {
                // the tile again, unscaled, and the regulariser by comparisons (rare)
#pragma unroll
                for (int it = 0; it < NIT1; ++it)
                {
                    const int idx = tid + it * kThreads;
                    const int r = idx / RV, c4 = idx - r * RV;
                    const int gi = i0 - HALO + r, gj = j0 - HL + 4 * c4;
                    const bool in = gi >= 0 && gi < H && gj >= 0 && gj < W;
                    if (idx < NCH1)
                        *(float4*)&tile[4 * idx] = in ? stage[it] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                __syncthreads();
                if (c + 1 < CH)
                    load_stage(c + 1);
                lds_cfloat* tc = (lds_cfloat*)tile + lrow * PITCH + 4 * tx + HL;
#pragma unroll
                for (int e = 0; e < PXT; ++e)
                    centre[c][e] = tc[e];
                lds_cfloat* wl = (lds_cfloat*)whole_weights;
                float reg[PXT] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
                for (int m = 0; m <= 3; ++m)
                {
#pragma unroll 1
                    for (int l = 3; l + m >= 0; --l)
                    {
                        const float wt = *wl++;
#pragma unroll
                        for (int e = 0; e < PXT; ++e)
                            reg[e] += wt * (dsign(centre[c][e], tc[m * PITCH + e + l]) - dsign(tc[-m * PITCH + e - l], centre[c][e]));
                    }
                }
                reg01 = v2f{reg[0], reg[1]}, reg23 = v2f{reg[2], reg[3]};
            }
            {
                const float reg[PXT] = {reg01.x, reg01.y, reg23.x, reg23.y};
#pragma unroll
                for (int e = 0; e < PXT; ++e)
                {
                    if (c < kPark)
                        park[(c * PXT + e) * kThreads + tid] = reg[e];
                    else
                        keep[c][e] = reg[e];
                }
            }
            __syncthreads(); // the tile is rewritten for the next channel
            SRX_STAMP(2 + c);
        }
    }
    else
        __syncthreads(); // the weight table
    // ---- scale 2: the patch of each frame's sign plane that this tile back-projects from goes through LDS
    // (k_tile_relative); the headers of all frames are fetched once, one per lane
    static_assert(FOOT != 3 || (TYN == kUpdTileRows && TW == kUpdTileCols && kThreads == 256), "tile geometry of k_tile_relative");
    __shared__ uint32_t sgl[FOOT == 3 ? 2 * kSgWords : 1];
    uint32_t hdrs = 0, s_g = 0, s_l = 0;
    uint32_t sign_last = 0;
    if (FOOT == 3)
    {
        for (int k = 0; k < K; ++k)
        {
            const uint32_t v = ((const uint32_t*)dec_fwd.p[k])[(size_t)H * W + t];
            hdrs = (tid & 63) == k ? v : hdrs; // every lane loads the same word
        }
        // a thread stages column tid & 63 of rows (tid >> 6) + 4 * it, it = 0..2, of the patch
        s_g = (uint32_t)((tid >> 6) * (W >> 1) + (tid & 63)) * 4u;
        s_l = (uint32_t)((tid >> 6) * kSgPitch + (tid & 63));
        sign_last = (uint32_t)(((H >> 1) + 1) * (W >> 1) - 1) * 4u; // last cell of the spare row
    }
    static_assert(kSgRows * kSgCols == 3 * 256 && kSgCols == 64, "three cells per thread, one patch row per wave");
    // cells beyond the plane are never referenced by an entry; their loads are bent to the plane's last cell
    auto stage_load = [&](int k, uint32_t hd, uint32_t (&v)[3]) {
        const char* base = (const char*)signs.p[k];
        const uint32_t org = (hd & kFwdCellMask) + s_g;
        const uint32_t four_rows = (uint32_t)(W >> 1) * 16u;
#pragma unroll
        for (int it = 0; it < 3; ++it)
            v[it] = *(const uint32_t*)(base + min(org + it * four_rows, sign_last));
    };
    auto stage_put = [&](int buf, const uint32_t (&v)[3]) {
#pragma unroll
        for (int it = 0; it < 3; ++it)
            sgl[buf * kSgWords + it * 4 * kSgPitch + s_l] = v[it];
    };
    // ---- sum over frames of M^T H^T D^T sign, in units of 1/256
    int acc[PXT][CH];
    int bias[PXT];
#pragma unroll
    for (int p = 0; p < PXT; ++p)
    {
        bias[p] = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            acc[p][c] = 0;
    }
    const char* lutb = (const char*)lut;
    {
        uint32_t accv[PXT];
#pragma unroll
        for (int p = 0; p < PXT; ++p)
            accv[p] = 0;
        int pending = 0;
        uint32_t hd_now = 0; // scale 2: header of the current frame's tile (wave-uniform)
        // scale 4: the four pixels' 2x2 blocks of sign cells of one frame times the entries' weights
        auto gather4 = [&](const char* sg, const uint32_t (&ev)[PXT]) {
            uint4_a8 qs[PXT];
#pragma unroll
            for (int p = 0; p < PXT; ++p)
                qs[p] = *(const uint4_a8*)(sg + (ev[p] & kFwdCellMask));
            __builtin_amdgcn_sched_barrier(0); // all four gathers are in flight before the first is consumed
#pragma unroll
            for (int p = 0; p < PXT; ++p)
            {
                const uint32_t e = ev[p];
                const uint4_a8 q = qs[p];
                const char* le = lutb + ((e >> (kFwdLutShift - 4)) & 0x1F0u);
                const uint4 wq = *(const uint4*)le;
                uint32_t v = accv[p];
                v = mad_u24(q.x, wq.x, v);
                v = mad_u24(q.y, wq.y, v);
                v = mad_u24(q.z, wq.z, v);
                v = mad_u24(q.w, wq.w, v);
                accv[p] = v;
                bias[p] += *(const uint32_t*)(le + kFwdLutSize * 16);
            }
        };
        // A frame without motion (scale 4): its forward map is the identity, so its entries follow from the
        // pixels' own place (decode_fwd of (j + p, i)) and are not read.  Integer sums: the order of the frames
        // is free, this one goes first, while the first frame's entries are still on their way.
        if (FOOT == 2 && still_signs)
        {
            const int py = i & 3;
            const uint32_t cell = (uint32_t)(((i >> 2) + (py == 3)) * (W >> 2) + (j >> 2)) * 8u;
            const uint32_t ly = (uint32_t)(5 * (i == H - 2 ? 4 : py)) << kFwdLutShift;
            const bool last_row = i >= H - 1;
            uint32_t ev[PXT];
#pragma unroll
            for (int p = 0; p < PXT; ++p)
            {
                const uint32_t lx = (uint32_t)((p == 2 && j == W - 4) ? 4 : p) << kFwdLutShift;
                const bool none = last_row || (p == 3 && j == W - 4); // decode_fwd: targets in the last row / column
                ev[p] = none ? kFwdLutZero << kFwdLutShift : (cell + (p == 3 ? 8u : 0u)) | (lx + ly);
            }
            gather4((const char*)still_signs, ev);
            pending = 1;
        }
        // one window frame: gathers through the entries in `now` while the next frame's entries arrive in `next`
        auto frame = [&](int k, const uint4& now, uint4& next) {
            if (k + 1 < K)
            {
                if (FOOT == 3 && ((uint32_t)__builtin_amdgcn_readlane((int)hdrs, k + 1) & kSgFit))
                {
                    const uint2 n16 = stream_load((const uint2*)((const uint32_t*)dec_fwd.p[k + 1] + short_entries) + pix / 4);
                    next = make_uint4(n16.x, n16.y, 0, 0);
                }
                else
                    next = stream_load((const uint4*)((const uint32_t*)dec_fwd.p[k + 1] + pix));
            }
            const char* sg = (const char*)signs.p[k];
            const uint32_t ev[PXT] = {now.x, now.y, now.z, now.w};
            if (FOOT == 2)
                gather4(sg, ev);
            else if (hd_now & kSgFit)
            {
                // the tile's patch of this frame is in LDS: nine cells per pixel from there
                const uint32_t* lb = sgl + (k & 1) * kSgWords;
#pragma unroll
                for (int p = 0; p < PXT; ++p)
                {
                    const uint32_t e = ev[p >> 1] >> (16 * (p & 1)); // 16-bit entries, two to a word
                    const uint32_t* q = lb + (e & kSgCellMask);
                    const char* le = lutb + ((e >> kSgLutShift) & 31u) * 48u;
                    const uint4 w0 = *(const uint4*)le, w1 = *(const uint4*)(le + 16), w2 = *(const uint4*)(le + 32);
                    uint32_t v = accv[p];
                    v = mad_u24(q[0], w0.x, v);
                    v = mad_u24(q[1], w0.y, v);
                    v = mad_u24(q[2], w0.z, v);
                    v = mad_u24(q[kSgPitch], w0.w, v);
                    v = mad_u24(q[kSgPitch + 1], w1.x, v);
                    v = mad_u24(q[kSgPitch + 2], w1.y, v);
                    v = mad_u24(q[2 * kSgPitch], w1.z, v);
                    v = mad_u24(q[2 * kSgPitch + 1], w1.w, v);
                    v = mad_u24(q[2 * kSgPitch + 2], w2.x, v);
                    accv[p] = v;
                    bias[p] += w2.y;
                    if (SRX_B3_LDS_PX == 1 || (p & 1))
                        __builtin_amdgcn_sched_barrier(0); // bounds the registers held by reads in flight
                }
            }
            else
            {
                // (a tile whose entries did not fit the patch) three rows of three cells per pixel from global
                // memory, two pixels at a time (register budget)
                const size_t row_bytes = (size_t)(W >> 1) * 4;
#pragma unroll
                for (int p0 = 0; p0 < PXT; p0 += 2)
                {
                    uint3_a4 qs[2][3];
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int r = 0; r < 3; ++r)
                            qs[p][r] = *(const uint3_a4*)(sg + (ev[p0 + p] & kFwdCellMask) + r * row_bytes);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                    {
                        const uint32_t e = ev[p0 + p];
                        const char* le = lutb + ((e >> kFwdLutShift) & 31u) * 48u;
                        const uint4 w0 = *(const uint4*)le, w1 = *(const uint4*)(le + 16), w2 = *(const uint4*)(le + 32);
                        uint32_t v = accv[p0 + p];
                        v = mad_u24(qs[p][0].x, w0.x, v);
                        v = mad_u24(qs[p][0].y, w0.y, v);
                        v = mad_u24(qs[p][0].z, w0.z, v);
                        v = mad_u24(qs[p][1].x, w0.w, v);
                        v = mad_u24(qs[p][1].y, w1.x, v);
                        v = mad_u24(qs[p][1].z, w1.y, v);
                        v = mad_u24(qs[p][2].x, w1.z, v);
                        v = mad_u24(qs[p][2].y, w1.w, v);
                        v = mad_u24(qs[p][2].z, w2.x, v);
                        accv[p0 + p] = v;
                        bias[p0 + p] += w2.y;
                    }
                }
            }
            if (++pending == fw.flush || k + 1 == K)
            {
#pragma unroll
                for (int p = 0; p < PXT; ++p)
                {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        acc[p][c] += (int)((accv[p] >> (10 * c)) & 1023u);
                    accv[p] = 0;
                }
                pending = 0;
            }
        };
        // scale 2: the patch of frame k+2 is asked for while frame k is worked on (two frames ahead: one frame's
        // arithmetic is shorter than a memory latency), frame k+1's is written to the other LDS buffer behind
        // frame k's reads; one barrier per frame.  `par` = k & 1.
        uint32_t sv[SRX_B3_DEPTH][3];
        auto step = [&](int k, const uint4& now, uint4& next, const int par) {
            if (FOOT == 3)
            {
                const uint32_t hd_next = k + 1 < K ? (uint32_t)__builtin_amdgcn_readlane((int)hdrs, k + 1) : 0u;
                if (SRX_B3_DEPTH == 2)
                {
                    const uint32_t hd_far = k + 2 < K ? (uint32_t)__builtin_amdgcn_readlane((int)hdrs, k + 2) : 0u;
                    if (hd_far & kSgFit)
                        stage_load(k + 2, hd_far, sv[par]);
                }
                else if (hd_next & kSgFit)
                    stage_load(k + 1, hd_next, sv[0]);
                frame(k, now, next);
                if (hd_next & kSgFit)
                    stage_put(par ^ 1, sv[SRX_B3_DEPTH == 2 ? par ^ 1 : 0]);
                __syncthreads();
                hd_now = hd_next;
            }
            else
                frame(k, now, next);
        };
        if (FOOT == 3 && K > 0)
        {
            hd_now = (uint32_t)__builtin_amdgcn_readlane((int)hdrs, 0);
            if (hd_now & kSgFit)
                cur = make_uint4(cur16.x, cur16.y, 0, 0);
            const uint32_t hd_1 = K > 1 ? (uint32_t)__builtin_amdgcn_readlane((int)hdrs, 1) : 0u;
            if (hd_now & kSgFit)
                stage_load(0, hd_now, sv[0]);
            if (SRX_B3_DEPTH == 2 && (hd_1 & kSgFit))
                stage_load(1, hd_1, sv[SRX_B3_DEPTH - 1]);
            if (hd_now & kSgFit)
                stage_put(0, sv[0]);
            __syncthreads();
        }
        // two frames per trip, the two entry registers swapping roles (no copies)
        uint4 alt = cur;
        int k = 0;
        for (; k + 1 < K; k += 2)
        {
            step(k, cur, alt, 0);
            step(k + 1, alt, cur, 1);
        }
        if (k < K)
            step(k, cur, alt, 0);
        if (pending) // (only the frame without motion was there)
        {
#pragma unroll
            for (int p = 0; p < PXT; ++p)
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    acc[p][c] += (int)((accv[p] >> (10 * c)) & 1023u);
        }
    }
    SRX_STAMP(5);
    // ---- update
    if (!live)
        return;
#pragma unroll
    for (int c = 0; c < CH; ++c)
    {
        // the regulariser comes from its parking slot
        // with the regulariser on, the centre pixels were kept from the tile (-0 arrives as +0, which the
        // double sum below maps to the same result); otherwise they are read here
        float cxv[PXT];
        if (lambda > 0 && !(FOOT == 3 && SRX_B3_REREAD))
        {
#pragma unroll
            for (int p = 0; p < PXT; ++p)
                cxv[p] = centre[c][p];
        }
        else
        {
            const float4 ctr = *(const float4*)(X + c * ps + pix);
            cxv[0] = ctr.x, cxv[1] = ctr.y, cxv[2] = ctr.z, cxv[3] = ctr.w;
        }
        float outv[PXT];
#pragma unroll
        for (int p = 0; p < PXT; ++p)
        {
            const bool interior = i >= HALO && i < H - HALO && (j + p) >= HALO && (j + p) < W - HALO;
            float d = (float)(acc[p][c] - bias[p]) * 0.00390625f;
            if (lambda > 0)
            {
                const float rv = c < kPark ? park[(c * PXT + p) * kThreads + tid] : keep[c][p];
                const float rg = interior ? rv : 0.0f;
                d = (float)((double)d + (double)rg * (-lambda));
            }
            outv[p] = (float)((double)cxv[p] + (double)d * tau);
        }
        *(float4*)(X_out + c * ps + pix) = make_float4(outv[0], outv[1], outv[2], outv[3]);
    }
    SRX_STAMP(6);
}

// plain sign plane <-> pair plane (stage-level entry points)
__global__ void k_sign_pairs(const uint32_t* __restrict__ plain, uint2* __restrict__ pairs, int h, int w)
{
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)h * w;
    if (o > n)
        return;
    uint2 v = make_uint2(0, 0); // the spare cell and the row below the last are zero
    if (o < n)
    {
        v.x = plain[o];
        if (o + w < n)
            v.y = plain[o + w];
    }
    pairs[o] = v;
}
__global__ void k_sign_unpair(const uint2* __restrict__ pairs, uint32_t* __restrict__ plain, size_t n)
{
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o < n)
        plain[o] = pairs[o].x;
}

// ---------------------------------------------------------------------------
// ps == 0: interleaved source; otherwise planar with plane stride ps
// DEPTH: 0 = 8-bit, 2 = 16-bit (convertTo: cvRound, half to even, then clamp; NaN -> 0), 5 = float
template <int CH, int DEPTH>
__global__ void k_crop_convert(const float* __restrict__ X, int rows, int cols, int border, void* __restrict__ dst,
                               size_t ps)
{
    const int oc = cols - 2 * border, orr = rows - 2 * border;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= oc || y >= orr)
        return;
    const size_t pi = (size_t)(y + border) * cols + (x + border);
    const size_t o = ((size_t)y * oc + x) * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c)
    {
        const float v = X[ps ? c * ps + pi : pi * CH + c];
        if (DEPTH == 0)
        {
            float r = rintf(v);
            r = fminf(fmaxf(r, 0.0f), 255.0f);
            ((uint8_t*)dst)[o + c] = (uint8_t)(int)r;
        }
        else if (DEPTH == 2)
        {
            float r = rintf(v);
            r = fminf(fmaxf(r, 0.0f), 65535.0f);
            ((uint16_t*)dst)[o + c] = (uint16_t)(int)r;
        }
        else
            ((float*)dst)[o + c] = v;
    }
}

template <typename T>
__global__ void k_convert_to_32f(const T* __restrict__ src, size_t count, float* __restrict__ dst)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count)
        dst[e] = (float)src[e];
}

inline dim3 grid2d(int cols, int rows, dim3 b, int z = 1)
{
    return dim3((cols + b.x - 1) / b.x, (rows + b.y - 1) / b.y, z);
}

int check_launch(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return hip_fail(e, what);
    return 0;
}

// round the block count up to a multiple of 8 so that every XCD's range exists
inline unsigned xcd_grid(unsigned n)
{
    return (n + 7u) & ~7u;
}

} // namespace

// ---------------------------------------------------------------------------
int launch_relative_motions(const PtrList& fwd, const MutPtrList& rel, int n, int base_idx, size_t count,
                            hipStream_t st)
{
    const int bs = 256;
    hipLaunchKernelGGL(k_relative_motions, dim3((unsigned)((count + bs - 1) / bs)), dim3(bs), 0, st, fwd, rel, n,
                       base_idx, count);
    return check_launch("relative_motions");
}

static int launch_group_maps(const int16_t* bwd_map, uint32_t* dec_bwd, size_t hr_px, int hr_cols, hipStream_t st)
{
    const int bs = 256;
    const size_t entries = hr_px / 4; // scale 4: every row is a whole number of groups
    hipLaunchKernelGGL(k_group_maps, dim3((unsigned)((entries + bs - 1) / bs)), dim3(bs), 0, st,
                       (const short2*)bwd_map, dec_bwd, (int)(hr_px / hr_cols), hr_cols);
    if (int rc = check_launch("group_maps"))
        return rc;
    const int hr_rows = (int)(hr_px / hr_cols), w = hr_cols / 4, h = hr_rows / 4;
    const int strips_x = (w + kFwdStrip - 1) / kFwdStrip;
    hipLaunchKernelGGL(k_group_flags, dim3(strips_x, h), dim3(64), 0, st, dec_bwd, (uint8_t*)(dec_bwd + 3 * (hr_px / 4)),
                       hr_rows, hr_cols, strips_x);
    return check_launch("group_flags");
}

// scale 2: entries relative to the update kernel's tiles, headers behind the map (dec_fwd_words)
static int launch_tile_relative(uint32_t* dec_fwd, int hr_rows, int hr_cols, hipStream_t st)
{
    if (hr_cols % 4 != 0 || hr_cols < 8 || hr_rows < 8)
        return 0; // the specialised update kernel does not run at such sizes (backproject_fast_eligible)
    const int tiles_x = (hr_cols + kUpdTileCols - 1) / kUpdTileCols, tiles_y = (hr_rows + kUpdTileRows - 1) / kUpdTileRows;
    hipLaunchKernelGGL(k_tile_relative, dim3((unsigned)(tiles_x * tiles_y)), dim3(256), 0, st, dec_fwd, hr_rows, hr_cols,
                       tiles_x, (unsigned)dec_fwd_tile_words(hr_rows, hr_cols));
    return check_launch("tile_relative");
}

int launch_build_maps(const float* rel, int rows, int cols, int scale, const CubicTables& t, int16_t* fwd_map,
                      int16_t* bwd_map, uint32_t* dec_fwd, uint32_t* dec_bwd, hipStream_t st)
{
    if ((dec_bwd && scale != 4) || (dec_fwd && scale != 4 && scale != 2))
    {
        set_error("build_maps: decoded maps exist for scale 4 (both) and scale 2 (forward) only");
        return 1;
    }
    if (scale == 4 && dec_fwd && dec_bwd && g_kernel_path == 0)
    {
        // raw maps, decoded forward map, group map and row kinds in one pass
        const int strips_x = (cols + kFwdStrip - 1) / kFwdStrip;
        const size_t hr_px = (size_t)rows * 4 * cols * 4;
        hipLaunchKernelGGL(k_build_maps_cells, dim3(strips_x, rows + 1), dim3(64), 0, st, rel, rows, cols, t,
                           (short2*)fwd_map, (short2*)bwd_map, dec_fwd, dec_bwd, (uint8_t*)(dec_bwd + 3 * (hr_px / 4)),
                           strips_x);
        return check_launch("build_maps_cells");
    }
    const dim3 b(64, 4);
    hipLaunchKernelGGL(k_build_maps, grid2d(cols * scale, rows * scale, b), b, 0, st, rel, rows, cols, scale, t,
                       (short2*)fwd_map, (short2*)bwd_map, dec_fwd);
    if (int rc = check_launch("build_maps"))
        return rc;
    if (dec_fwd && scale == 2)
        if (int rc = launch_tile_relative(dec_fwd, rows * scale, cols * scale, st))
            return rc;
    return dec_bwd ? launch_group_maps(bwd_map, dec_bwd, (size_t)rows * scale * cols * scale, cols * scale, st) : 0;
}

int launch_decode_maps(const int16_t* fwd_map, const int16_t* bwd_map, int hr_rows, int hr_cols, int scale,
                       uint32_t* dec_fwd, uint32_t* dec_bwd, hipStream_t st)
{
    const dim3 b(64, 4);
    if (dec_fwd)
    {
        hipLaunchKernelGGL(k_decode_maps, grid2d(hr_cols, hr_rows, b), b, 0, st, (const short2*)fwd_map, hr_rows,
                           hr_cols, scale, dec_fwd);
        if (int rc = check_launch("decode_maps"))
            return rc;
        if (scale == 2)
            if (int rc = launch_tile_relative(dec_fwd, hr_rows, hr_cols, st))
                return rc;
    }
    return dec_bwd ? launch_group_maps(bwd_map, dec_bwd, (size_t)hr_rows * hr_cols, hr_cols, st) : 0;
}

int launch_resize_cubic(const float* src, int srows, int scols, int ch, const CubicTables& t, float* dst, int drows,
                        int dcols, size_t plane_stride, hipStream_t st)
{
    const dim3 b(64, 4);
    if (plane_stride != 0 && plane_stride % 4 == 0 && drows == 4 * srows && dcols == 4 * scols && (ch == 1 || ch == 3) &&
        g_kernel_path == 0)
    {
        const dim3 gc = grid2d(scols, srows, b);
        if (ch == 1)
            hipLaunchKernelGGL(k_resize_cubic_cells<1>, gc, b, 0, st, src, srows, scols, t, dst, plane_stride);
        else
            hipLaunchKernelGGL(k_resize_cubic_cells<3>, gc, b, 0, st, src, srows, scols, t, dst, plane_stride);
        return check_launch("resize_cubic_cells");
    }
    const dim3 g = grid2d(dcols, drows, b);
    switch (ch)
    {
    case 1: hipLaunchKernelGGL(k_resize_cubic<1>, g, b, 0, st, src, srows, scols, t, dst, drows, dcols, plane_stride); break;
    case 2: hipLaunchKernelGGL(k_resize_cubic<2>, g, b, 0, st, src, srows, scols, t, dst, drows, dcols, plane_stride); break;
    case 3: hipLaunchKernelGGL(k_resize_cubic<3>, g, b, 0, st, src, srows, scols, t, dst, drows, dcols, plane_stride); break;
    case 4: hipLaunchKernelGGL(k_resize_cubic<4>, g, b, 0, st, src, srows, scols, t, dst, drows, dcols, plane_stride); break;
    default: set_error("resize_cubic: channels must be 1..4"); return 1;
    }
    return check_launch("resize_cubic");
}

int launch_relayout(const float* src, float* dst, size_t pixels, int ch, size_t plane_stride, bool to_planar,
                    hipStream_t st)
{
    const int bs = 256;
    const dim3 g((unsigned)((pixels + bs - 1) / bs));
    if (ch == 1)
        hipLaunchKernelGGL(k_relayout<1>, g, dim3(bs), 0, st, src, dst, pixels, plane_stride, to_planar ? 1 : 0);
    else if (ch == 3)
        hipLaunchKernelGGL(k_relayout<3>, g, dim3(bs), 0, st, src, dst, pixels, plane_stride, to_planar ? 1 : 0);
    else
    {
        set_error("relayout: channels must be 1 or 3");
        return 1;
    }
    return check_launch("relayout");
}

int launch_btv_regularization(const float* src, int rows, int cols, int ch, const BtvW& w, float* dst, hipStream_t st)
{
    const dim3 b(64, 4);
    const dim3 g = grid2d(cols, rows, b);
    switch (ch)
    {
    case 1: hipLaunchKernelGGL(k_btv_regularization<1>, g, b, 0, st, src, rows, cols, w, dst); break;
    case 3: hipLaunchKernelGGL(k_btv_regularization<3>, g, b, 0, st, src, rows, cols, w, dst); break;
    default: set_error("btv_regularization: channels must be 1 or 3"); return 1;
    }
    return check_launch("btv_regularization");
}

int launch_sign_pairs(const uint32_t* plain, uint32_t* pairs, int lr_rows, int lr_cols, hipStream_t st)
{
    const int bs = 256;
    const size_t n = (size_t)lr_rows * lr_cols + 1;
    hipLaunchKernelGGL(k_sign_pairs, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, st, plain, (uint2*)pairs,
                       lr_rows, lr_cols);
    return check_launch("sign_pairs");
}

int launch_sign_unpair(const uint32_t* pairs, uint32_t* plain, int lr_rows, int lr_cols, hipStream_t st)
{
    const int bs = 256;
    const size_t n = (size_t)lr_rows * lr_cols;
    hipLaunchKernelGGL(k_sign_unpair, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, st, (const uint2*)pairs, plain,
                       n);
    return check_launch("sign_unpair");
}

bool forward_fast_eligible(int hr_rows, int hr_cols, int ch, int scale, const GaussK& g)
{
    // (the kernel reaches X and the group map through buffer descriptors: 31-bit byte offsets)
    const size_t hr_px = (size_t)hr_rows * hr_cols;
    return scale == 4 && g.n == 5 && (ch == 1 || ch == 3) && hr_rows % 4 == 0 && hr_cols % 4 == 0 &&
           hr_rows <= kGrpMaxDim && hr_cols <= kGrpMaxDim && (size_t)ch * (hr_px + 4) * sizeof(float) < 0x7FFFFFF0u;
}

// A launch of one of the two solver kernels.  With g_launch_timer armed (Solver::run, kernel timing) the launch goes
// through hipExtLaunchKernelGGL, which ties the two events to the dispatch's own start and end: their distance is
// the kernel's execution time, as a profiler reports it.  (Events recorded in front of and behind a launch also
// count the time the queue takes to get from the marker to the kernel.)  The first launch after arming takes the
// start event, every launch moves the stop event.
thread_local LaunchTimer g_launch_timer;
#define SRX_LAUNCH_TIMED(kernel, grid, block, lds, st, ...)                                                            \
    do                                                                                                                 \
    {                                                                                                                  \
        if (g_launch_timer.stop)                                                                                       \
        {                                                                                                              \
            hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), st, g_launch_timer.start,                 \
                                  g_launch_timer.stop, 0, __VA_ARGS__);                                                \
            g_launch_timer.start = nullptr;                                                                            \
        }                                                                                                              \
        else                                                                                                           \
            hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                             \
    } while (0)

// CUs of the current device and, on first use per device, the forward kernel's dynamic-LDS limit (a per-device
// attribute); 0 with the error set
static int forward_device_setup()
{
    constexpr int kMaxDevices = 64;
    static std::atomic<int> cus_of[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
    {
        check_launch("forward_sign_lds (device query)");
        return 0;
    }
    int cus = cus_of[dev].load(std::memory_order_acquire);
    if (cus == 0)
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess)
        {
            check_launch("forward_sign_lds (device query)");
            return 0;
        }
        if (hipFuncSetAttribute((const void*)k_forward_sign_lds<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)forward_lds_bytes(1)) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_forward_sign_lds<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)forward_lds_bytes(3)) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_forward_sign_lds<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)forward_lds_bytes(1)) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_forward_sign_lds<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)forward_lds_bytes(3)) != hipSuccess)
        {
            check_launch("forward_sign_lds (LDS size)");
            return 0;
        }
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        cus_of[dev].store(cus, std::memory_order_release);
    }
    return cus;
}

// Tiles of one launch of k_forward_sign_lds: as many as one round of resident blocks holds
static int forward_tile_slots(int ch, int K, int cus)
{
    const int passes = (K + kFwdMaxWaves - 1) / kFwdMaxWaves;
    const int blocks_per_cu = forward_lds_bytes(ch) * 2 <= 160 * 1024 ? 2 : 1;
    return std::max(1, blocks_per_cu * cus / passes);
}

// Cost of a row served from the window, surcharge per frame with a mixed row that leaves the window and per frame
// with an outside row, the most all surcharges of a row add up to, bound on the tiles per strip (rows / min_rows);
// tuning aid:
// SRX_FWD_PLAN="inside,mixed,outside,most_extra,min_rows" in the environment replaces them (read once; inside = 0
// turns the plan off)
static PlanCost g_plan_cost = [] {
    // (measured on clips with exact and with TV-L1 motion: a row costs about twice as much once any frame's wave
    // leaves the window, and hardly more when several do)
    // (small images: down to two rows per tile - 320x240 runs 5 % faster per iteration than with at least four)
    PlanCost pc{16, 16, 16, 16, 2};
    if (const char* e = std::getenv("SRX_FWD_PLAN"))
    {
        int v[5];
        if (std::sscanf(e, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]) == 5 && v[0] >= 0 && v[4] >= 1)
            pc = PlanCost{v[0], v[1], v[2], v[3], v[4]};
    }
    return pc;
}();

size_t forward_plan_bytes(int hr_rows, int hr_cols, int ch, int K, size_t* scratch_bytes)
{
    const int w = hr_cols / 4, h = hr_rows / 4;
    const int strips_x = (w + kFwdStrip - 1) / kFwdStrip;
    if (scratch_bytes)
        *scratch_bytes = (size_t)h * strips_x * sizeof(int);
    // room for any device this build runs on (the tile count follows the CU count)
    return (size_t)std::max(forward_tile_slots(ch, K, 1024), strips_x) * sizeof(int4);
}

int launch_plan_forward_tiles(const PtrList& dec_bwd, int K, int hr_rows, int hr_cols, int ch, void* plan, void* scratch,
                              bool* planned, hipStream_t st)
{
    *planned = false;
    const int w = hr_cols / 4;
    const int strips_x = (w + kFwdStrip - 1) / kFwdStrip;
    const int cus = forward_device_setup();
    if (cus == 0)
        return 1;
    const int n_tiles = forward_tile_slots(ch, K, cus);
    if (g_plan_cost.inside <= 0 || strips_x > n_tiles || strips_x > kPlanMaxStrips || K < 1 || K > kMaxWindow)
        return 0; // the even cut
    hipLaunchKernelGGL(k_plan_tiles, dim3(1), dim3(kPlanThreads), 0, st, dec_bwd, K, hr_rows, hr_cols, strips_x, n_tiles,
                       g_plan_cost, (int*)scratch, (int4*)plan);
    *planned = true;
    return check_launch("plan_tiles");
}

int launch_forward_sign(const float* X, int hr_rows, int hr_cols, int ch, const PtrList& raw_bwd,
                        const PtrList* dec_bwd, const PtrList& y, const MutPtrList& signs, int K, int scale,
                        const GaussK& g, size_t generic_plane_stride, hipStream_t st, const PtrList* y_bytes, int still_frame,
                        const void* tile_plan)
{
    if (ch != 1 && ch != 3)
    {
        set_error("forward_sign: channels must be 1 or 3");
        return 1;
    }
    if (dec_bwd && forward_fast_eligible(hr_rows, hr_cols, ch, scale, g))
    {
        const int w = hr_cols / 4, h = hr_rows / 4;
        const size_t ps = plane_stride_for((size_t)hr_rows * hr_cols);
        const int strips_x = (w + kFwdStrip - 1) / kFwdStrip;
        // one wave per window frame, at most kFwdMaxWaves frames per block (more frames: more passes)
        const int passes = (K + kFwdMaxWaves - 1) / kFwdMaxWaves;
        const int waves = (K + passes - 1) / passes;
        // cut the rows so that one round of resident blocks covers the image (evenly, or as planned)
        const int cus = forward_device_setup();
        if (cus == 0)
            return 1;
        const int slots_all = forward_tile_slots(ch, K, cus);
        const int slots = std::max(1, slots_all / strips_x);
        const int rows_per_tile = std::max(4, (h + slots - 1) / slots);
        const int row_groups = (h + rows_per_tile - 1) / rows_per_tile;
        const int4* plan = (const int4*)tile_plan;
        const unsigned n_tiles = plan ? (unsigned)slots_all : (unsigned)strips_x * row_groups;
        const dim3 gr(xcd_grid(n_tiles), passes), b(64 * waves);
        const size_t lds = forward_lds_bytes(ch);
        // the still frame goes to the last wave of the (single) pass
        const int shift = passes == 1 && still_frame >= 0 && still_frame < K ? (still_frame + 1) % K : 0;
        if (y_bytes)
        {
            if (ch == 1)
                SRX_LAUNCH_TIMED((k_forward_sign_lds<1, true>), gr, b, lds, st, X, hr_rows, hr_cols, ps, raw_bwd, *dec_bwd,
                                   *y_bytes, signs, K, g, strips_x, rows_per_tile, n_tiles, shift, plan);
            else
                SRX_LAUNCH_TIMED((k_forward_sign_lds<3, true>), gr, b, lds, st, X, hr_rows, hr_cols, ps, raw_bwd, *dec_bwd,
                                   *y_bytes, signs, K, g, strips_x, rows_per_tile, n_tiles, shift, plan);
        }
        else if (ch == 1)
            SRX_LAUNCH_TIMED((k_forward_sign_lds<1, false>), gr, b, lds, st, X, hr_rows, hr_cols, ps, raw_bwd, *dec_bwd, y,
                               signs, K, g, strips_x, rows_per_tile, n_tiles, shift, plan);
        else
            SRX_LAUNCH_TIMED((k_forward_sign_lds<3, false>), gr, b, lds, st, X, hr_rows, hr_cols, ps, raw_bwd, *dec_bwd, y,
                               signs, K, g, strips_x, rows_per_tile, n_tiles, shift, plan);
        return check_launch("forward_sign_lds");
    }
    if (g_kernel_path == 0 && scale == 2 && g.n == 5 && generic_plane_stride != 0 && hr_cols >= 8 && hr_rows >= 8 &&
        hr_rows % 2 == 0 && hr_cols % 2 == 0 && (size_t)hr_rows * hr_cols < (1u << 31))
    {
        const int w = hr_cols / 2, h = hr_rows / 2;
        const int row_groups = (h + kFwd2Rows - 1) / kFwd2Rows;
        if (hr_cols % 4 == 0 && g_forward_s2_wide)
        {
            const int strips_x = (w + kFwd2Cols - 1) / kFwd2Cols;
            const unsigned n_items = (unsigned)strips_x * row_groups * K;
            const dim3 gr(xcd_grid(n_items)), b(64);
#define SRX_S2W(CHN, BYTES, YS) SRX_LAUNCH_TIMED((k_forward_sign_s2w<CHN, BYTES>), gr, b, 0, st, X, hr_rows, hr_cols, generic_plane_stride, raw_bwd, YS, signs, K, g, strips_x, n_items)
            if (y_bytes && ch == 1)
                SRX_S2W(1, true, *y_bytes);
            else if (y_bytes)
                SRX_S2W(3, true, *y_bytes);
            else if (ch == 1)
                SRX_S2W(1, false, y);
            else
                SRX_S2W(3, false, y);
#undef SRX_S2W
            return check_launch("forward_sign_s2w");
        }
        const int strips_x = (w + kFwd2Strip - 1) / kFwd2Strip;
        const unsigned n_items = (unsigned)strips_x * row_groups * K;
        const dim3 gr(xcd_grid(n_items)), b(64);
        if (y_bytes)
        {
            if (ch == 1)
                SRX_LAUNCH_TIMED((k_forward_sign_s2<1, true>), gr, b, 0, st, X, hr_rows, hr_cols, generic_plane_stride,
                                   raw_bwd, *y_bytes, signs, K, g, strips_x, n_items);
            else
                SRX_LAUNCH_TIMED((k_forward_sign_s2<3, true>), gr, b, 0, st, X, hr_rows, hr_cols, generic_plane_stride,
                                   raw_bwd, *y_bytes, signs, K, g, strips_x, n_items);
        }
        else if (ch == 1)
            SRX_LAUNCH_TIMED((k_forward_sign_s2<1, false>), gr, b, 0, st, X, hr_rows, hr_cols, generic_plane_stride,
                               raw_bwd, y, signs, K, g, strips_x, n_items);
        else
            SRX_LAUNCH_TIMED((k_forward_sign_s2<3, false>), gr, b, 0, st, X, hr_rows, hr_cols, generic_plane_stride,
                               raw_bwd, y, signs, K, g, strips_x, n_items);
        return check_launch("forward_sign_s2");
    }
    const size_t tile_lds = forward_tile_lds_bytes(ch, scale, g.n);
    if (g_kernel_path == 0 && tile_lds <= 64 * 1024 && (size_t)hr_rows * hr_cols < (1u << 31))
    {
        const int w = hr_cols / scale, h = hr_rows / scale;
        const dim3 gr((w + kTileLrW - 1) / kTileLrW, (h + kTileLrH - 1) / kTileLrH, K), b(256);
#define SRX_TILE(CHN, FIVE) SRX_LAUNCH_TIMED((k_forward_sign_tile<CHN, FIVE>), gr, b, tile_lds, st, X, hr_rows, hr_cols, generic_plane_stride, raw_bwd, y, signs, scale, g)
        if (ch == 1 && g.n == 5)
            SRX_TILE(1, true);
        else if (ch == 1)
            SRX_TILE(1, false);
        else if (g.n == 5)
            SRX_TILE(3, true);
        else
            SRX_TILE(3, false);
#undef SRX_TILE
        return check_launch("forward_sign_tile");
    }
    const dim3 b(64, 4);
    const dim3 gr = grid2d(hr_cols / scale, hr_rows / scale, b, K);
    if (ch == 1)
        SRX_LAUNCH_TIMED(k_forward_sign<1>, gr, b, 0, st, X, hr_rows, hr_cols, generic_plane_stride, raw_bwd, y, signs,
                           scale, g);
    else
        SRX_LAUNCH_TIMED(k_forward_sign<3>, gr, b, 0, st, X, hr_rows, hr_cols, generic_plane_stride, raw_bwd, y, signs,
                           scale, g);
    return check_launch("forward_sign");
}

// The integer path needs every blur weight to be a multiple of 1/16 (true for
// the sigma = 0 table {1,4,6,4,1}/16).
static bool make_fast_weights(const GaussK& g, int scale, FastW& fw)
{
    if (g.n != 5 || (scale != 4 && scale != 2))
        return false;
    int ki[3];
    for (int i = 0; i < 3; ++i)
    {
        const float v = g.k[2 + i] * 16.0f;
        if (!(v >= 0.0f && v <= 15.0f) || v != (float)(int)v || g.k[2 + i] != g.k[2 - i])
            return false;
        ki[i] = (int)v;
    }
    std::memset(&fw, 0, sizeof(fw));
    uint32_t wmax = 0;
    if (scale == 4)
    {
        // per phase p: slot A carries tap j = {0,-1,-2,+1,-2}[p], slot B only exists for p == 2 (tap +2)
        const int gA[5] = {ki[0], ki[1], ki[2], ki[1], ki[2]};
        const int gB[5] = {0, 0, ki[2], 0, 0};
        for (int py = 0; py < 5; ++py)
            for (int px = 0; px < 5; ++px)
            {
                const int p = px + 5 * py;
                fw.w[p][0] = gA[px] * gA[py]; // (row, col)
                fw.w[p][1] = gA[px] * gB[py]; // (row+1, col)
                fw.w[p][2] = gB[px] * gA[py]; // (row, col+1)
                fw.w[p][3] = gB[px] * gB[py]; // (row+1, col+1)
            }
    }
    else
    {
        // per phase (decode_fwd2) the weights of the three cells from the base cell on
        const int ax[5][3] = {{ki[2], ki[0], ki[2]}, {ki[1], ki[1], 0}, {ki[0], 2 * ki[2], 0}, {2 * ki[1], 0, 0},
                              {ki[2], ki[0] + ki[2], 0}};
        for (int py = 0; py < 5; ++py)
            for (int px = 0; px < 5; ++px)
                for (int dy = 0; dy < 3; ++dy)
                    for (int dx = 0; dx < 3; ++dx)
                        fw.w[px + 5 * py][3 * dy + dx] = ax[px][dx] * ax[py][dy];
    }
    for (int p = 0; p < 25; ++p)
    {
        for (int i = 0; i < 9; ++i)
            fw.wsum[p] += fw.w[p][i];
        wmax = fw.wsum[p] > wmax ? fw.wsum[p] : wmax;
    }
    if (wmax == 0 || 2 * wmax > 1023u)
        return false; // 10-bit fields
    fw.flush = (int)(1023u / (2 * wmax));
    return true;
}

// Scale 4 reads pair planes (the specialised forward kernel writes them), scale 2 plain planes with one
// spare row behind the last (cells of weight 0 are still read).
bool backproject_fast_eligible(int hr_rows, int hr_cols, int ch, int scale, const GaussK& g, const BtvW& w)
{
    FastW fw;
    if (w.radius != 3 || (ch != 1 && ch != 3) || hr_rows % 4 != 0 || hr_cols % 4 != 0 || !make_fast_weights(g, scale, fw))
        return false;
    for (int i = 0; i < 22; ++i) // the kernel works with twice the weights: doubling must be exact
        if (!(std::fabs(w.w[i]) < 0x1p100f) || (w.w[i] * 2.0f) * 0.5f != w.w[i])
            return false;
    if (scale == 4)
        return ((size_t)(hr_rows / 4) * (hr_cols / 4) + 1) * 8 <= (1u << 24);
    return hr_rows >= 8 && hr_cols >= 8 && ((size_t)(hr_rows / 2 + 1) * (hr_cols / 2) + 4) * 4 <= (1u << 24);
}

int launch_backproject_update(const float* X, float* X_out, int hr_rows, int hr_cols, int ch, const PtrList& raw_fwd,
                              const PtrList* dec_fwd, const PtrList& signs, int K, int scale, const GaussK& g,
                              const BtvW& w, double tau, double lambda, hipStream_t st, int identity_frame)
{
    if (ch != 1 && ch != 3)
    {
        set_error("backproject_update: channels must be 1 or 3");
        return 1;
    }
    FastW fw;
    if (dec_fwd && backproject_fast_eligible(hr_rows, hr_cols, ch, scale, g, w) && make_fast_weights(g, scale, fw))
    {
        constexpr int tile_rows = SRX_B_THREADS / 16;
        const int tiles_x = (hr_cols + 63) / 64, tiles_y = (hr_rows + tile_rows - 1) / tile_rows;
        const size_t ps = plane_stride_for((size_t)hr_rows * hr_cols);
        const unsigned n_tiles = (unsigned)tiles_x * tiles_y;
        const dim3 gr(xcd_grid(n_tiles)), b(SRX_B_THREADS);
        BtvW ws = w; // twice the weights (exact, checked by backproject_fast_eligible), see half_step2
        for (int i = 0; i < kMaxBtvWeights; ++i)
            ws.w[i] *= 2.0f;
        // scale 4: a frame vouched to have the identity as its forward map leaves the lists; the kernel works its
        // entries out by itself and is handed its sign plane separately
        PtrList maps = *dec_fwd, sgs = signs;
        const void* still = nullptr;
        int Kr = K;
        if (scale == 4 && identity_frame >= 0 && identity_frame < K)
        {
            still = signs.p[identity_frame];
            for (int k = identity_frame; k + 1 < K; ++k)
                maps.p[k] = dec_fwd->p[k + 1], sgs.p[k] = signs.p[k + 1];
            Kr = K - 1;
        }
#define SRX_UPD(CHN, FT) SRX_LAUNCH_TIMED((k_backproject_update_s4<CHN, FT>), gr, b, 0, st, X, X_out, hr_rows, hr_cols, ps, maps, sgs, Kr, fw, ws, tau, lambda, tiles_x, n_tiles, still)
        if (ch == 1 && scale == 4)
            SRX_UPD(1, 2);
        else if (ch == 1)
            SRX_UPD(1, 3);
        else if (scale == 4)
            SRX_UPD(3, 2);
        else
            SRX_UPD(3, 3);
#undef SRX_UPD
        return check_launch("backproject_update_s4");
    }
    const dim3 b(64, 4);
    const dim3 gr = grid2d(hr_cols, hr_rows, b);
    if (ch == 1)
        SRX_LAUNCH_TIMED(k_backproject_update<1>, gr, b, 0, st, X, X_out, hr_rows, hr_cols, raw_fwd, signs, K, scale,
                           g, w, tau, lambda);
    else
        SRX_LAUNCH_TIMED(k_backproject_update<3>, gr, b, 0, st, X, X_out, hr_rows, hr_cols, raw_fwd, signs, K, scale,
                           g, w, tau, lambda);
    return check_launch("backproject_update");
}

int launch_crop_convert(const float* X, int rows, int cols, int ch, int border, void* dst, int depth,
                        size_t plane_stride, hipStream_t st)
{
    const dim3 b(64, 4);
    const dim3 g = grid2d(cols - 2 * border, rows - 2 * border, b);
    if (depth != 0 && depth != 2 && depth != 5)
    {
        set_error("crop_convert: depth must be 8U, 16U or 32F");
        return 1;
    }
#define SRX_CROP(CHN, DEP) hipLaunchKernelGGL((k_crop_convert<CHN, DEP>), g, b, 0, st, X, rows, cols, border, dst, plane_stride)
    if (ch == 1)
    {
        if (depth == 0)
            SRX_CROP(1, 0);
        else if (depth == 2)
            SRX_CROP(1, 2);
        else
            SRX_CROP(1, 5);
    }
    else if (ch == 3)
    {
        if (depth == 0)
            SRX_CROP(3, 0);
        else if (depth == 2)
            SRX_CROP(3, 2);
        else
            SRX_CROP(3, 5);
    }
#undef SRX_CROP
    else
    {
        set_error("crop_convert: channels must be 1 or 3");
        return 1;
    }
    return check_launch("crop_convert");
}

int launch_convert_8u_to_32f(const uint8_t* src, size_t count, float* dst, hipStream_t st)
{
    const int bs = 256;
    hipLaunchKernelGGL(k_convert_to_32f<uint8_t>, dim3((unsigned)((count + bs - 1) / bs)), dim3(bs), 0, st, src, count,
                       dst);
    return check_launch("convert_8u_to_32f");
}

int launch_convert_16u_to_32f(const uint16_t* src, size_t count, float* dst, hipStream_t st)
{
    const int bs = 256;
    hipLaunchKernelGGL(k_convert_to_32f<uint16_t>, dim3((unsigned)((count + bs - 1) / bs)), dim3(bs), 0, st, src, count,
                       dst);
    return check_launch("convert_16u_to_32f");
}

} // namespace srx
