/*
 * srx — C ABI of the MI355X-native BTV-L1 multiframe super-resolution path.
 *
 * Every entry point is plain C (pointers, sizes, ints); no C++ or torch types
 * cross this boundary.  Each declaration cites the reference interface it
 * replaces (paths relative to the reference tree).  The C++ facade in
 * include/srx/superres.hpp re-creates the reference's classes on top of it.
 *
 * Threading: a handle is a single stream-ordered context bound to one GPU; it
 * is not thread-safe (neither is the reference object).  Handles on different
 * GPUs may be driven from different threads.
 *
 * Errors: functions return SRX_OK (0) or an SRX_ERR_* code; srx_last_error()
 * returns the calling thread's last message.
 */
#ifndef SRX_H
#define SRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRX_API __attribute__((visibility("default")))

enum
{
    SRX_OK = 0,
    SRX_ERR_INVALID = 1, /* bad argument or parameter value            */
    SRX_ERR_STATE = 2,   /* call not valid in the handle's current state */
    SRX_ERR_DEVICE = 3,  /* a HIP call failed                          */
    SRX_ERR_SOURCE = 4,  /* the frame source ran dry during warm-up (CV_Assert at lib/src/btv_l1.cpp:479) */
    SRX_ERR_FLOW = 5     /* the optical-flow plug-in failed             */
};

/* depth codes are OpenCV's CV_8U / CV_16U / CV_32F.  Frames convert to float unscaled
 * (frame.convertTo(CV_32F), lib/src/btv_l1.cpp:519); the flow stage rescales to 8 bit
 * (convertToType, lib/src/input_array_utility.cpp:90-103). */
enum
{
    SRX_8U = 0,
    SRX_16U = 2,
    SRX_32F = 5
};
enum
{
    SRX_MEM_HOST = 0,
    SRX_MEM_DEVICE = 1
};

/* Image view: the cv::Mat header reduced to what the path reads.  Channels
 * are interleaved; step is the row pitch in bytes (0 = dense). */
typedef struct srx_image
{
    void* data;
    int rows, cols, channels, depth;
    size_t step;
    int mem; /* SRX_MEM_HOST or SRX_MEM_DEVICE */
} srx_image;

typedef struct srx_handle srx_handle;

/* videostab::IFrameSource as used by lib/include/super_resolution.hpp:149
 * (shape visible in samples/quality_test/main.cpp:43-50): next() fills *out
 * and returns 1, or returns 0 at the end of the stream (the empty Mat). The
 * image must stay valid until the next call of next() (the library has finished
 * reading it - also from device or pinned memory - before it calls next() again). */
typedef int (*srx_next_frame_fn)(void* user, srx_image* out);
typedef void (*srx_reset_fn)(void* user);

/* DenseOpticalFlow::calc(frame0, frame1, flow1) of
 * lib/include/optical_flow.hpp:44 for user plug-ins: frames are handed over as
 * the source delivered them; the plug-in fills *flow (rows x cols, 2 channels,
 * SRX_32F, interleaved dx,dy; host or device) and returns 0 on success. */
typedef int (*srx_flow_fn)(void* user, const srx_image* frame0, const srx_image* frame1, srx_image* flow);
/* The two-output form calc(frame0, frame1, flow1, flow2) of the same interface
 * (lib/include/optical_flow.hpp:44, lib/src/optical_flow.cpp:71-81): the plug-in fills two planes, flow_u and
 * flow_v (rows x cols, 1 channel, SRX_32F each).
 * Lifetime, both forms: the frames handed to the plug-in are valid during the call; the result images must stay
 * valid until the plug-in is called again on this handle or the call that invoked it returns, whichever comes
 * first (device and pinned results are read on the handle's stream, which the library waits for at those points). */
typedef int (*srx_flow2_fn)(void* user, const srx_image* frame0, const srx_image* frame1, srx_image* flow_u,
                            srx_image* flow_v);

enum
{
    SRX_FLOW_NONE = 0,    /* flow removed: streaming calls fail with SRX_ERR_FLOW */
    SRX_FLOW_TVL1 = 1,    /* built-in duality-based TV-L1 on the GPU             */
    SRX_FLOW_CALLBACK = 2, /* user plug-in                                       */
    /* built-in Farneback polynomial-expansion flow on the GPU: the reference's default flow class
     * (lib/src/btv_l1.cpp:287; parameters and defaults of lib/src/optical_flow.cpp:41-50) and the flow
     * of a new handle */
    SRX_FLOW_FARNEBACK = 3
};

/* ---- object life cycle ------------------------------------------------- */

/* `new BTV_L1` (samples/video/main.cpp:106); defaults of
 * BTV_L1_Base::BTV_L1_Base and BTV_L1::BTV_L1 (lib/src/btv_l1.cpp:277-295,
 * 461-464).  device >= 0 selects the GPU; device == -1 creates a dry-run
 * handle that runs the host-side bookkeeping only (no GPU needed, outputs
 * carry sizes but data == NULL). */
SRX_API int srx_create_btvl1(int device, srx_handle** out);
SRX_API void srx_destroy(srx_handle* h);
SRX_API const char* srx_last_error(void);

/* ---- parameters: Algorithm::set / getInt / getDouble -------------------- */
/* names and types as registered at lib/src/super_resolution.cpp:83-93:
 * scale, iterations, btvKernelSize, blurKernelSize, temporalAreaRadius (int);
 * tau, lambda, alpha, blurSigma (double).  Extra int names of this library:
 * outputMemory (SRX_MEM_*; default host), traceEvents (0/1), kernelTiming (0/1),
 * readAhead (0/1, default 0; built-in flow only): srx_next_frame pulls the frame of the NEXT call from the
 * source while the current solve runs and computes its flow on a second stream.  Same frames out; the
 * source is read one frame earlier than BTV_L1::processImpl (lib/src/btv_l1.cpp:490-509) would;
 * flowPairsInFlight (1 .. SRX_MAX_FLOW_PAIRS_IN_FLIGHT, default 3; built-in TV-L1 only): how many of the flows
 * frame i -> base that srx_process computes for a window (lib/src/btv_l1.cpp:311-321) run side by side. */
SRX_API int srx_set_int(srx_handle* h, const char* name, int value);
SRX_API int srx_set_double(srx_handle* h, const char* name, double value);
SRX_API int srx_get_int(srx_handle* h, const char* name, int* value);
SRX_API int srx_get_double(srx_handle* h, const char* name, double* value);

/* set("opticalFlow", Ptr<DenseOpticalFlow>) — lib/src/super_resolution.cpp:93,
 * samples/quality_test/main.cpp:267 */
SRX_API int srx_set_flow_builtin(srx_handle* h, int kind);
SRX_API int srx_set_flow_callback(srx_handle* h, srx_flow_fn fn, void* user);
SRX_API int srx_set_flow_callback2(srx_handle* h, srx_flow2_fn fn, void* user);
/* parameters of the built-in flows.  TV-L1: tau, lambda, theta, epsilon (double); nscales, warps,
 * iterations (int); iterationsPerLaunch (int: 0 = chosen by the size of the pyramid level, the default;
 * 1, 4, 8, 12 or 14 = one kernel at every level - the flows and iteration counts are the same).  Farneback (names of lib/include/optical_flow.hpp's class Farneback): pyrScale,
 * polySigma (double); numLevels, winSize, numIters, polyN, flags (int; 0 = box window, 256 = Gaussian
 * window as OPTFLOW_FARNEBACK_GAUSSIAN; an initial flow, flag 4, is not built). */
SRX_API int srx_flow_set_int(srx_handle* h, const char* name, int value);
SRX_API int srx_flow_set_double(srx_handle* h, const char* name, double value);
SRX_API int srx_flow_get_int(srx_handle* h, const char* name, int* value);
SRX_API int srx_flow_get_double(srx_handle* h, const char* name, double* value);

/* ---- streaming API: SuperResolution ------------------------------------- */

/* SuperResolution::setFrameSource, lib/src/super_resolution.cpp:131-135 */
SRX_API int srx_set_input(srx_handle* h, srx_next_frame_fn next, srx_reset_fn reset, void* user);
/* SuperResolution::reset, lib/src/super_resolution.cpp:137-141 */
SRX_API int srx_reset(srx_handle* h);
/* SuperResolution::nextFrame, lib/src/super_resolution.cpp:143-152 with
 * BTV_L1::initImpl / processImpl (lib/src/btv_l1.cpp:466-509).  *out
 * describes a buffer owned by the handle, valid until the next call on the
 * handle (the reference returns a Mat aliasing its member dst).  End of
 * stream: returns SRX_OK with out->rows == 0 (the empty Mat). */
SRX_API int srx_next_frame(srx_handle* h, srx_image* out);

/* Multi-GPU span support.  Restricts which outputs (indices counted from the
 * first frame the source delivers) are solved and returned; frames outside
 * the range still enter the ring and the flow stage so that windows at the
 * span edges see their halo.  first > last disables the restriction. */
SRX_API int srx_set_output_range(srx_handle* h, int first, int last);

/* ---- batch API: BTV_L1_Base::process ------------------------------------ */
/* lib/src/btv_l1.cpp:297-326 (motions == NULL: flow frame i -> base is
 * computed by the flow plug-in) and :328-355 (n-1 consecutive forward
 * motions).  frames: n images of equal size/type (8U or 32F, 1 or 3
 * channels).  out: caller buffer (host or device) of
 * (rows*scale-2B) x (cols*scale-2B), same depth as the frames.  A dry-run
 * handle (device -1) checks every argument and calls the flow plug-in as
 * often as a GPU handle would, writes nothing to out and returns SRX_OK. */
SRX_API int srx_process(srx_handle* h, const srx_image* frames, int n, const srx_image* motions, int base_idx,
                        srx_image* out);

/* DenseOpticalFlow::calc / collectGarbage, lib/include/optical_flow.hpp:44-45.
 * flow2 == NULL: one interleaved 2-channel result; otherwise planar u and v
 * (lib/src/optical_flow.cpp:71-81). */
SRX_API int srx_flow_calc(srx_handle* h, const srx_image* frame0, const srx_image* frame1, srx_image* flow1,
                          srx_image* flow2);
SRX_API int srx_collect_garbage(srx_handle* h);
/* The built-in flow (Farneback if that is the handle's flow, TV-L1 otherwise) on two
 * device-resident grey float images (rows x
 * cols, values 0..255) with the handle's flow parameters: flow = rows x cols
 * x 2 interleaved, device.  executed_iterations (optional) receives the
 * number of TV-L1 inner iterations that ran (0 for Farneback).  This is the standalone flow stage of
 * the path (the reference reaches its flow only through
 * DenseOpticalFlow::calc, lib/include/optical_flow.hpp:44). */
SRX_API int srx_flow_calc_device(srx_handle* h, const float* gray0, const float* gray1, int rows, int cols,
                                 float* flow, long long* executed_iterations);
/* The same for n independent pairs (gray0[i] -> gray1[i] into flows[i], all rows x cols, device), up to
 * in_flight of them side by side (1 .. SRX_MAX_FLOW_PAIRS_IN_FLIGHT), each on a stream and a TV-L1 workspace of
 * the handle's own: a pyramid's coarse levels are many short launches that leave most of the GPU idle, and
 * another pair's fill it.  Every flow is the one srx_flow_calc_device returns for its pair.
 * executed_iterations: n entries or NULL.  Returns when all flows are written.  Farneback pairs run one after the
 * other.  The reference computes the n - 1 motions of a window one DenseOpticalFlow::calc at a time
 * (lib/src/btv_l1.cpp:311-316); they are independent pairs like these. */
#define SRX_MAX_FLOW_PAIRS_IN_FLIGHT 4
SRX_API int srx_flow_calc_device_pairs(srx_handle* h, int n, const float* const* gray0, const float* const* gray1,
                                       int rows, int cols, float* const* flows, long long* executed_iterations,
                                       int in_flight);
/* Work of the handle's last built-in TV-L1 call (of all pairs of a srx_flow_calc_device_pairs call): pixels x executed inner iterations summed over pyramid
 * levels and warps (each inner iteration moves 64 bytes per pixel of its level; bench.py prices the flow
 * stage with it).  No counterpart in the reference. */
SRX_API int srx_flow_pixel_iterations(srx_handle* h, long long* pixel_iterations);

/* Event trace of the streaming driver (needs traceEvents = 1).  Each event is
 * 5 ints {type, a, b, c, d}: READ{got}, FLOW{i}, SOLVE{idx,start,end,base},
 * OUT{idx}.  Returns the number of events copied (oldest first) and clears
 * the log. */
enum
{
    SRX_EV_READ = 1,
    SRX_EV_FLOW = 2,
    SRX_EV_SOLVE = 3,
    SRX_EV_OUT = 4
};
SRX_API int srx_drain_events(srx_handle* h, int* buf, int max_events);

/* ---- pure host helpers (no GPU needed) ---------------------------------- */

/* at() of lib/src/ring_buffer.hpp:36-46: borderInterpolate(idx, size, BORDER_WRAP) */
SRX_API int srx_ring_slot(int index, int size);
/* window of BTV_L1::processFrame, lib/src/btv_l1.cpp:529-540 */
SRX_API void srx_window(int idx, int radius, int store_pos, int* start, int* end, int* base);
/* calcBtvWeights, lib/src/btv_l1.cpp:172-189; w must hold ksize*ksize floats;
 * returns the number of weights written (22 for ksize 7) */
SRX_API int srx_calc_btv_weights(int btv_kernel_size, double alpha, float* w);
/* getGaussianKernel as createGaussianFilter calls it, lib/src/btv_l1.cpp:394 */
SRX_API int srx_gaussian_kernel(int n, double sigma, float* k);
/* Contiguous span of output frames owned by `rank` of `world`, and the range
 * of input frames (halo of `radius` on both sides, clipped) it needs.  With world > 1 the plan is
 * refused (SRX_ERR_INVALID, for every rank alike) when some rank would be fed fewer than 2*radius+1
 * frames, which the warm-up of its stream needs (lib/src/btv_l1.cpp:476-481). */
SRX_API int srx_span_plan(int total_frames, int radius, int world, int rank, int* out_first, int* out_last,
                          int* in_first, int* in_last);

/* ---- stage-level device entry points ------------------------------------ */
/* All pointers are device pointers unless the name says host; `stream` is a
 * hipStream_t (NULL = default stream).  Images are dense, channels interleaved.
 * These are the kernels the solver is built from; they are exported so that
 * each stage can be checked in isolation. */

/* calcRelativeMotions, lib/src/btv_l1.cpp:36-57.  fwd_host/rel_host: host
 * arrays of device pointers (n-1 and n entries), each rows*cols*2 floats. */
SRX_API int srx_k_relative_motions(const float* const* fwd_host, float* const* rel_host, int n, int base_idx, int rows,
                                   int cols, void* stream);
/* upscaleMotions + buildMotionMaps + the float->int16 index conversion of
 * cv::remap(INTER_NEAREST): lib/src/btv_l1.cpp:59-101,428,441.  rel: LR
 * relative motion (rows x cols x 2); fwd_map / bwd_map: (rows*scale) x
 * (cols*scale) x 2 int16 source indices (x, y). */
SRX_API int srx_k_build_maps(const float* rel, int rows, int cols, int scale, int16_t* fwd_map, int16_t* bwd_map,
                             void* stream);
/* cv::resize(INTER_CUBIC) on float images, lib/src/btv_l1.cpp:75,412 */
SRX_API int srx_k_resize_cubic(const float* src, int srows, int scols, int ch, float* dst, int drows, int dcols,
                               void* stream);
/* calcBtvRegularization, lib/src/btv_l1.cpp:191-274 (weights from srx_calc_btv_weights) */
SRX_API int srx_k_btv_regularization(const float* src, int rows, int cols, int ch, int btv_kernel_size,
                                     const float* w_host, float* dst, void* stream);
/* One window frame of the forward chain, lib/src/btv_l1.cpp:428-434:
 * sign(y - D H M X).  X: hr_rows x hr_cols x ch; bwd_map int16 pairs;
 * y: (hr_rows/scale) x (hr_cols/scale) x ch; sign_out: one uint32 per LR
 * pixel holding sign+1 (0, 1 or 2) of channel c in bits 10c..10c+1. */
SRX_API int srx_k_forward_sign(const float* X, int hr_rows, int hr_cols, int ch, const int16_t* bwd_map,
                               const float* y, int scale, const float* gauss_host, int gauss_n, uint32_t* sign_out,
                               void* stream);
/* The transposed chain for K frames plus regulariser and update,
 * lib/src/btv_l1.cpp:437-452: X_out = X + tau * (sum_k M^T H^T D^T sign_k - lambda * btv(X)).
 * fwd_maps_host / signs_host: host arrays of K device pointers (sign planes
 * in the encoding of srx_k_forward_sign). */
SRX_API int srx_k_backproject_update(const float* X, float* X_out, int hr_rows, int hr_cols, int ch,
                                     const int16_t* const* fwd_maps_host, const uint32_t* const* signs_host, int K,
                                     int scale, const float* gauss_host, int gauss_n, int btv_kernel_size,
                                     const float* w_host, double tau, double lambda, void* stream);
/* Diagnostic switch, process-wide: 0 (default) lets the solver and the two
 * entry points above use the kernels specialised for scale 4 / 5-tap blur /
 * btvKernelSize 7 whenever the parameters allow; 1 forces the generic kernels.
 * Both compute the same values. */
SRX_API int srx_set_kernel_path(int path);
SRX_API int srx_get_kernel_path(void);
/* Rect inner(B,B,..) crop of lib/src/btv_l1.cpp:455-456 and, for depth
 * SRX_8U, Mat::convertTo(CV_8U) (lib/src/btv_l1.cpp:504). */
SRX_API int srx_k_crop_convert(const float* X, int rows, int cols, int ch, int border, void* dst, int depth,
                               void* stream);

/* Quality metrics of the reference's quality harness, computed on the GPU with
 * wavefront reductions.  Images: same size, depth and channel count (1..4),
 * host or device memory, any row pitch (a region of interest is passed as data
 * pointer + step, like goldFrame(inner) at samples/quality_test/main.cpp:320).
 * The calls return when the result is on the host; `stream` orders them after
 * earlier work on that stream.
 *  srx_quality_psnr   cv::PSNR as called at samples/quality_test/main.cpp:320,323
 *                     (8-bit images only): 20 log10(255 / (rms + DBL_EPSILON)).
 *  srx_quality_mssim  MSSIM() of samples/quality_test/main.cpp:140-199: 11x11
 *                     Gaussian (sigma 1.5) SSIM map, C1 = 6.5025, C2 = 58.5225,
 *                     averaged.  per_channel (4 doubles, may be NULL) receives
 *                     the mean of every channel's map; *mssim (may be NULL) the
 *                     value the reference's function returns: mean[0] for one
 *                     channel, else (mean[0] + mean[1] + mean[3]) / 3 -- the
 *                     reference reads component 3 where 2 is meant (:198). */
SRX_API int srx_quality_psnr(const srx_image* a, const srx_image* b, double* psnr, void* stream);
SRX_API int srx_quality_mssim(const srx_image* a, const srx_image* b, double* per_channel, double* mssim,
                              void* stream);

/* srx_quality_psnr for callers that cannot allocate device memory: code that sets
 * up or replays a captured HIP graph, tight serving loops, harnesses that forbid
 * hipMalloc.  It has no counterpart in the reference.  Same arguments and the same
 * result, bit for bit, as srx_quality_psnr, with these limits:
 *  - both images in device memory (SRX_MEM_DEVICE) on the current device;
 *  - at most 65536 row segments of 4096 bytes (rows * ceil(cols * channels /
 *    4096)), e.g. up to 65536 rows of 4096 bytes or 3840x2160 RGB (6480);
 *    larger pairs are refused with SRX_ERR_INVALID and a message.
 * The partial sums live in an array of 520 KiB that the library holds in every
 * GPU's memory from load to exit; the call makes no hipMalloc / hipFree and uses no
 * allocating library routine (two kernel launches, an 8-byte copy to the host, a
 * stream synchronise).  All calls on one GPU share that array: calls that overlap
 * in time on one GPU (other streams, other threads) race; order them.  Both
 * images must be plain device memory of the GPU that is current in the calling
 * thread (the stream is one of that GPU's): anything else is refused with
 * SRX_ERR_INVALID before a kernel is launched. */
SRX_API int srx_quality_psnr_noalloc(const srx_image* a, const srx_image* b, double* psnr, void* stream);

/* BTV_L1_Base::run, lib/src/btv_l1.cpp:357-457, on device-resident inputs
 * with the handle's parameters.  y_host / rel_host: host arrays of K device
 * pointers (float LR frames rows x cols x ch; relative motions rows x cols x
 * 2).  out: device float (rows*scale-2B) x (cols*scale-2B) x ch; x_full:
 * optional device float HR image before the crop (may be NULL). */
SRX_API int srx_run(srx_handle* h, const float* const* y_host, const float* const* rel_host, int K, int base_idx,
                    int rows, int cols, int ch, float* out, float* x_full);
/* same, from n-1 consecutive forward motions: lib/src/btv_l1.cpp:328-355 */
SRX_API int srx_process_device(srx_handle* h, const float* const* y_host, const float* const* fwd_motions_host, int K,
                               int base_idx, int rows, int cols, int ch, float* out, float* x_full);
/* Per-kernel timing of the solver loop, measured with HIP events on the
 * handle's stream while the int parameter "kernelTiming" is 1 (setting it also
 * clears the totals): summed durations of the forward (sign) and
 * back-projection/update launches and the number of iterations they cover
 * (every eighth iteration of a solve is timed, with events tied to the
 * kernels' own begin and end).
 * Synchronises the stream.  No counterpart in the reference (its samples time
 * whole frames, samples/video/main.cpp:40-47). */
SRX_API int srx_kernel_times(srx_handle* h, double* forward_ms, double* backproject_ms, long* iterations);
/* dst = src, 16 bytes per lane, bytes a multiple of 16 (device pointers, 16-byte aligned): the plain streaming
 * copy whose rate bench.py sets beside the kernels' as the measured ceiling of the memory system.  No
 * counterpart in the reference. */
SRX_API int srx_k_copy(const void* src, void* dst, size_t bytes, void* stream);
/* Stream the handle launches on (hipStream_t), for callers that time or order work against it. */
SRX_API void* srx_stream(srx_handle* h);
/* Blocks until everything queued on the handle's stream has finished. */
SRX_API int srx_synchronize(srx_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* SRX_H */
