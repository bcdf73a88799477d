// C ABI of include/srx.h: handle, parameter table, streaming driver
// (BTV_L1::initImpl / processImpl / addNewFrame / processFrame) and the
// stage-level entry points.
#include "srx_host.h"
#include "srx_tvl1.h"
#include "srx_farneback.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>

using namespace srx;

namespace
{

size_t elem_size(int depth)
{
    return depth == SRX_8U ? 1 : (depth == SRX_16U ? 2 : 4);
}

int check_image(const srx_image& im, const char* what)
{
    if (!im.data || im.rows < 1 || im.cols < 1 || (im.depth != SRX_8U && im.depth != SRX_16U && im.depth != SRX_32F) || im.channels < 1 ||
        im.channels > 4 || (im.mem != SRX_MEM_HOST && im.mem != SRX_MEM_DEVICE))
    {
        set_error(std::string(what) + ": malformed image header");
        return SRX_ERR_INVALID;
    }
    const size_t rowbytes = (size_t)im.cols * im.channels * elem_size(im.depth);
    if (im.step != 0 && im.step < rowbytes)
    {
        set_error(std::string(what) + ": step smaller than a row");
        return SRX_ERR_INVALID;
    }
    return SRX_OK;
}

// dense copy of an image (host or device, any pitch) into device memory
int copy_to_device(const srx_image& im, void* dst, hipStream_t st)
{
    const size_t rowbytes = (size_t)im.cols * im.channels * elem_size(im.depth);
    const size_t step = im.step ? im.step : rowbytes;
    SRX_HIP(copy_pitched(dst, rowbytes, im.data, step, rowbytes, im.rows,
                         im.mem == SRX_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
    return SRX_OK;
}

// dense device image -> caller image (host or device, any pitch)
int copy_from_device(const void* src, const srx_image& im, hipStream_t st)
{
    const size_t rowbytes = (size_t)im.cols * im.channels * elem_size(im.depth);
    const size_t step = im.step ? im.step : rowbytes;
    SRX_HIP(copy_pitched(im.data, step, src, rowbytes, rowbytes, im.rows,
                         im.mem == SRX_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
    return SRX_OK;
}

struct Event
{
    int v[5];
};

} // namespace

struct srx_handle
{
    int device = -1;
    bool dry = false;
    hipStream_t stream = nullptr;
    Params p;

    // optical flow plug-in
    int flow_kind = SRX_FLOW_FARNEBACK; // opticalFlow = new Farneback, lib/src/btv_l1.cpp:287
    srx_flow_fn flow_fn = nullptr;
    srx_flow2_fn flow2_fn = nullptr; // two-output form: planar u, v
    void* flow_user = nullptr;
    Tvl1Params flow_params;
    Tvl1 tvl1;
    FarnebackParams fb_params;
    Farneback farneback;
    // srx_flow_calc_device_pairs: independent pairs in flight side by side, each on a lane (a stream and a TV-L1
    // workspace of its own); the first lane is `stream` and `tvl1`
    struct FlowLane
    {
        Tvl1 tvl1;
        hipStream_t stream = nullptr;
    };
    std::vector<std::unique_ptr<FlowLane>> flow_lanes;
    long long pairs_pixel_iterations = -1; // summed over the pairs of the last ..._pairs call, else -1
    // the built-in flow of this handle on two frames, on the given stream
    int builtin_flow(const srx_image& f0, const srx_image& f1, float* dst, hipStream_t st)
    {
        if (flow_kind == SRX_FLOW_FARNEBACK)
            return farneback.calc_images(f0, f1, fb_params, dst, st);
        return tvl1.calc_images(f0, f1, flow_params, dst, nullptr, st);
    }

    // frame source
    srx_next_frame_fn next = nullptr;
    srx_reset_fn reset = nullptr;
    void* user = nullptr;
    bool firstCall = true;

    // rings (lib/include/super_resolution.hpp:182-190)
    std::vector<DeviceBuffer> frames, motions, results;
    std::vector<DeviceBuffer> frames8; // the 8-bit originals of `frames`, kept when the source delivers 8-bit frames
    int storePos = -1, procPos = -1, outPos = -1;
    int radius = 0; // temporalAreaRadius latched at initImpl (ring sizes are fixed there)
    int rows = 0, cols = 0, ch = 0, depth = SRX_8U;

    // prevFrame, kept the way the source delivered it
    std::vector<uint8_t> prev_host;
    DeviceBuffer prev_dev;
    srx_image prev_hdr{};
    DeviceBuffer staging, flow_dev, flow_planes, batch_out;
    std::vector<DeviceBuffer> batch_frames, batch_motions, batch_frames8;
    PinnedBuffer dst_host;
    int outputMemory = SRX_MEM_HOST;

    // readAhead: the frame of the NEXT call is read, and its flow computed on a second stream,
    // while the solve of this call runs.  Same results, one frame more read from the source.
    int readAhead = 0;
    int flowPairsInFlight = 3; // srx_process without motions: built-in TV-L1 flows of a window side by side
    hipStream_t stream2 = nullptr;
    hipEvent_t prev_ready = nullptr;
    bool look_valid = false, look_end = false;
    std::vector<uint8_t> look_host;
    DeviceBuffer look_dev, look_flow;
    srx_image look_hdr{};

    // Reads of caller memory (source frames, flow plug-in results) are queued on the stream; the caller may
    // reuse that memory at the next call-back, so the last queued read is marked by an event and waited for
    // before the source or the plug-in is called again.
    hipEvent_t caller_read = nullptr;
    bool caller_read_pending = false;
    int note_caller_reads()
    {
        if (dry)
            return SRX_OK;
        if (!caller_read)
            SRX_HIP(hipEventCreateWithFlags(&caller_read, hipEventDisableTiming));
        SRX_HIP(hipEventRecord(caller_read, stream));
        caller_read_pending = true;
        return SRX_OK;
    }
    int wait_caller_reads()
    {
        if (caller_read_pending)
        {
            SRX_HIP(hipEventSynchronize(caller_read));
            caller_read_pending = false;
        }
        return SRX_OK;
    }
    int pull_frame(srx_image* f, int* got)
    {
        if (int rc = wait_caller_reads())
            return rc;
        *got = next(user, f);
        log(SRX_EV_READ, *got ? 1 : 0);
        return SRX_OK;
    }

    int out_first = 0, out_last = -1; // first > last: unrestricted
    bool trace = false;
    std::vector<Event> events;

    Solver solver;

    void log(int type, int a = 0, int b = 0, int c = 0, int d = 0)
    {
        if (trace)
            events.push_back(Event{{type, a, b, c, d}});
    }
    bool in_range(int idx) const { return out_first > out_last || (idx >= out_first && idx <= out_last); }
    int slot(int idx) const { return srx_ring_slot(idx, 2 * radius + 1); }
    int set_device() const
    {
        if (!dry)
            SRX_HIP(hipSetDevice(device));
        return SRX_OK;
    }

    int to_float(const srx_image& im, float* dst, DeviceBuffer* keep_bytes = nullptr);
    int add_new_frame(const srx_image& frame, bool flow_ready = false);
    int read_ahead();
    int process_frame(int idx);
    int init_impl();
    int process_impl(srx_image* out, bool* produced);
    int calc_flow(const srx_image& f0, const srx_image& f1, float* dst_dev);
};

// frame.convertTo(at(storePos, frames), CV_32F): lib/src/btv_l1.cpp:519
// keep_bytes: where an 8-bit frame's bytes stay on the device (the forward kernel reads them in place of the floats)
int srx_handle::to_float(const srx_image& im, float* dst, DeviceBuffer* keep_bytes)
{
    const size_t count = (size_t)im.rows * im.cols * im.channels;
    if (im.depth == SRX_32F)
        return copy_to_device(im, dst, stream);
    DeviceBuffer& raw = im.depth == SRX_8U && keep_bytes ? *keep_bytes : staging;
    if (int rc = raw.reserve(count * elem_size(im.depth)))
        return rc;
    if (int rc = copy_to_device(im, raw.get(), stream))
        return rc;
    if (im.depth == SRX_16U)
        return launch_convert_16u_to_32f(raw.as<uint16_t>(), count, dst, stream);
    return launch_convert_8u_to_32f(raw.as<uint8_t>(), count, dst, stream);
}

// opticalFlow->calc(frame0, frame1, flow): lib/src/btv_l1.cpp:315,522
int srx_handle::calc_flow(const srx_image& f0, const srx_image& f1, float* dst)
{
    const size_t bytes = (size_t)f0.rows * f0.cols * 2 * sizeof(float);
    if (flow_kind == SRX_FLOW_CALLBACK)
    {
        if (int rc = wait_caller_reads()) // the plug-in may reuse the buffer of its previous result
            return rc;
        if (flow2_fn)
        {
            // calc(frame0, frame1, flow1, flow2): planar u and v (lib/src/optical_flow.cpp:71-81)
            srx_image fu{}, fv{};
            if (flow2_fn(flow_user, &f0, &f1, &fu, &fv) != 0)
            {
                set_error("optical-flow plug-in reported failure");
                return SRX_ERR_FLOW;
            }
            if (dry)
                return SRX_OK;
            const size_t px = (size_t)f0.rows * f0.cols;
            for (const srx_image* f : {&fu, &fv})
            {
                if (int rc = check_image(*f, "flow plug-in result"))
                    return rc;
                if (f->rows != f0.rows || f->cols != f0.cols || f->channels != 1 || f->depth != SRX_32F)
                {
                    set_error("flow plug-in results must be rows x cols x 1 float each");
                    return SRX_ERR_FLOW;
                }
            }
            if (int rc = flow_planes.reserve(px * 2 * sizeof(float)))
                return rc;
            float* u = flow_planes.as<float>();
            if (int rc = copy_to_device(fu, u, stream))
                return rc;
            if (int rc = copy_to_device(fv, u + px, stream))
                return rc;
            if (int rc = launch_merge2(u, u + px, px, dst, stream))
                return rc;
            return note_caller_reads();
        }
        srx_image fl{};
        if (flow_fn(flow_user, &f0, &f1, &fl) != 0)
        {
            set_error("optical-flow plug-in reported failure");
            return SRX_ERR_FLOW;
        }
        if (dry)
            return SRX_OK;
        if (int rc = check_image(fl, "flow plug-in result"))
            return rc;
        if (fl.rows != f0.rows || fl.cols != f0.cols || fl.channels != 2 || fl.depth != SRX_32F)
        {
            set_error("flow plug-in result must be rows x cols x 2 float");
            return SRX_ERR_FLOW;
        }
        (void)bytes;
        if (int rc = copy_to_device(fl, dst, stream))
            return rc;
        return note_caller_reads();
    }
    if (flow_kind == SRX_FLOW_TVL1 || flow_kind == SRX_FLOW_FARNEBACK)
    {
        if (dry)
            return SRX_OK;
        return builtin_flow(f0, f1, dst, stream);
    }
    set_error("no optical flow set (srx_set_flow_builtin / srx_set_flow_callback)");
    return SRX_ERR_FLOW;
}

// BTV_L1::addNewFrame, lib/src/btv_l1.cpp:511-525
// With flow_ready the motion prevFrame -> frame is already in look_flow (read_ahead).
int srx_handle::add_new_frame(const srx_image& frame, bool flow_ready)
{
    if (int rc = check_image(frame, "frame source"))
        return rc;
    if (frame.channels != 1 && frame.channels != 3)
    {
        set_error("frames must have 1 or 3 channels");
        return SRX_ERR_INVALID;
    }
    if (storePos < 0)
        rows = frame.rows, cols = frame.cols, ch = frame.channels, depth = frame.depth;
    else if (frame.rows != rows || frame.cols != cols || frame.channels != ch || frame.depth != depth)
    {
        set_error("frame size or type changed within the stream");
        return SRX_ERR_INVALID;
    }
    ++storePos;
    const size_t px = (size_t)rows * cols;
    if (!dry)
    {
        DeviceBuffer& f = frames[slot(storePos)];
        if (int rc = f.reserve(px * ch * sizeof(float)))
            return rc;
        if (int rc = to_float(frame, f.as<float>(), &frames8[slot(storePos)]))
            return rc;
    }
    if (storePos > 0 && flow_ready)
    {
        if (!dry)
            std::swap(motions[slot(storePos - 1)], look_flow);
    }
    else if (storePos > 0)
    {
        log(SRX_EV_FLOW, storePos - 1);
        float* dst = nullptr;
        if (!dry)
        {
            DeviceBuffer& m = motions[slot(storePos - 1)];
            if (int rc = m.reserve(px * 2 * sizeof(float)))
                return rc;
            dst = m.as<float>();
        }
        if (int rc = calc_flow(prev_hdr, frame, dst))
            return rc;
    }
    // frame.copyTo(prevFrame)
    const size_t rowbytes = (size_t)cols * ch * elem_size(depth);
    prev_hdr = frame;
    prev_hdr.step = rowbytes;
    if (frame.mem == SRX_MEM_HOST)
    {
        prev_host.resize(rowbytes * rows);
        const size_t step = frame.step ? frame.step : rowbytes;
        for (int y = 0; y < rows; ++y)
            std::memcpy(prev_host.data() + (size_t)y * rowbytes, (const uint8_t*)frame.data + (size_t)y * step,
                        rowbytes);
        prev_hdr.data = prev_host.data();
    }
    else if (!dry)
    {
        if (int rc = prev_dev.reserve(rowbytes * rows))
            return rc;
        if (int rc = copy_to_device(frame, prev_dev.get(), stream))
            return rc;
        prev_hdr.data = prev_dev.get();
    }
    return note_caller_reads(); // the frame's memory is the source's again at its next call
}

// Reads the next call's frame now and computes its flow on the second stream; the solve enqueued
// on the main stream runs meanwhile (the flow's many small launches fit beside it).
int srx_handle::read_ahead()
{
    srx_image f{};
    int got = 0;
    if (int rc = pull_frame(&f, &got))
        return rc;
    if (!got)
    {
        look_end = true;
        return SRX_OK;
    }
    if (int rc = check_image(f, "frame source"))
        return rc;
    if (f.rows != rows || f.cols != cols || f.channels != ch || f.depth != depth)
    {
        set_error("frame size or type changed within the stream");
        return SRX_ERR_INVALID;
    }
    const size_t rowbytes = (size_t)cols * ch * elem_size(depth);
    look_hdr = f;
    look_hdr.step = rowbytes;
    if (!dry && !stream2)
    {
        SRX_HIP(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
        SRX_HIP(hipEventCreateWithFlags(&prev_ready, hipEventDisableTiming));
    }
    if (f.mem == SRX_MEM_HOST)
    {
        look_host.resize(rowbytes * rows);
        const size_t step = f.step ? f.step : rowbytes;
        for (int y = 0; y < rows; ++y)
            std::memcpy(look_host.data() + (size_t)y * rowbytes, (const uint8_t*)f.data + (size_t)y * step, rowbytes);
        look_hdr.data = look_host.data();
    }
    else if (!dry)
    {
        if (int rc = look_dev.reserve(rowbytes * rows))
            return rc;
        // the main stream may still be reading look_dev (the frame of this call) and writing prevFrame
        SRX_HIP(hipStreamWaitEvent(stream2, prev_ready, 0));
        if (int rc = copy_to_device(f, look_dev.get(), stream2))
            return rc;
        look_hdr.data = look_dev.get();
    }
    log(SRX_EV_FLOW, storePos);
    if (!dry)
    {
        if (int rc = look_flow.reserve((size_t)rows * cols * 2 * sizeof(float)))
            return rc;
        SRX_HIP(hipStreamWaitEvent(stream2, prev_ready, 0)); // prevFrame was written on the main stream
        if (int rc = builtin_flow(prev_hdr, look_hdr, look_flow.as<float>(), stream2))
            return rc;
        SRX_HIP(hipStreamSynchronize(stream2));
    }
    look_valid = true;
    return SRX_OK;
}

// BTV_L1::processFrame, lib/src/btv_l1.cpp:527-549
int srx_handle::process_frame(int idx)
{
    if (!in_range(idx))
        return SRX_OK;
    int start, end, base;
    srx_window(idx, radius, storePos, &start, &end, &base);
    log(SRX_EV_SOLVE, idx, start, end, base);
    if (dry)
        return SRX_OK;
    const float* y[kMaxWindow];
    const float* fm[kMaxWindow];
    const uint8_t* y8[kMaxWindow];
    const int K = end - start + 1;
    for (int i = start, k = 0; i <= end; ++i, ++k)
    {
        y[k] = frames[slot(i)].as<float>();
        y8[k] = frames8[slot(i)].as<uint8_t>();
        if (i < end)
            fm[k] = motions[slot(i)].as<float>();
    }
    if (depth == SRX_8U)
        solver.byte_frames(y8, K);
    if (int rc = solver.process(y, fm, K, base, rows, cols, ch, p, nullptr, nullptr, stream))
        return rc;
    // results ring keeps the cropped frame already converted to 8 bit; the
    // reference converts at output time (lib/src/btv_l1.cpp:504), same values
    const int R = rows * p.scale, C = cols * p.scale, B = p.btvKernelSize;
    DeviceBuffer& res = results[slot(idx)];
    if (int rc = res.reserve((size_t)(R - 2 * B) * (C - 2 * B) * ch))
        return rc;
    return launch_crop_convert(solver.last_x(), R, C, ch, B, res.get(), SRX_8U, solver.last_x_plane_stride(), stream);
}

// BTV_L1::initImpl, lib/src/btv_l1.cpp:466-488
int srx_handle::init_impl()
{
    if (int rc = validate_params(p))
        return rc;
    radius = p.temporalAreaRadius;
    const int cache = 2 * radius + 1;
    frames.clear();
    results.clear();
    motions.clear();
    frames.resize(cache);
    frames8.clear();
    frames8.resize(cache);
    results.resize(cache);
    motions.resize(cache);
    storePos = -1;
    for (int t = -radius; t <= radius; ++t)
    {
        srx_image frame{};
        int got = 0;
        if (int rc = pull_frame(&frame, &got))
            return rc;
        if (!got)
        {
            set_error("frame source ended during warm-up (needs 2*temporalAreaRadius+1 frames)");
            return SRX_ERR_SOURCE;
        }
        if (int rc = add_new_frame(frame))
            return rc;
    }
    for (int i = 0; i <= radius; ++i)
        if (int rc = process_frame(i))
            return rc;
    procPos = radius;
    outPos = -1;
    return SRX_OK;
}

// BTV_L1::processImpl, lib/src/btv_l1.cpp:490-509
int srx_handle::process_impl(srx_image* out, bool* produced)
{
    *produced = false;
    srx_image frame{};
    int got = 0;
    const bool ahead = look_valid;
    if (ahead)
        frame = look_hdr, got = 1, look_valid = false;
    else if (!look_end)
    {
        if (int rc = pull_frame(&frame, &got))
            return rc;
    }
    if (got)
        if (int rc = add_new_frame(frame, ahead))
            return rc;
    const bool look = readAhead && (flow_kind == SRX_FLOW_TVL1 || flow_kind == SRX_FLOW_FARNEBACK) && got && !look_end;
    if (look && !dry && stream2)
        SRX_HIP(hipEventRecord(prev_ready, stream));
    if (procPos < storePos)
    {
        ++procPos;
        if (int rc = process_frame(procPos))
            return rc;
    }
    if (look)
    {
        if (!dry && !stream2)
        {
            SRX_HIP(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
            SRX_HIP(hipEventCreateWithFlags(&prev_ready, hipEventDisableTiming));
            // first use: the copy of prevFrame is somewhere in the work already enqueued
            SRX_HIP(hipEventRecord(prev_ready, stream));
        }
        if (int rc = read_ahead())
            return rc;
    }
    if (outPos < storePos)
    {
        ++outPos;
        *produced = true;
        if (!in_range(outPos))
            return SRX_OK;
        log(SRX_EV_OUT, outPos);
        const int B = p.btvKernelSize;
        out->rows = rows * p.scale - 2 * B;
        out->cols = cols * p.scale - 2 * B;
        out->channels = ch;
        out->depth = SRX_8U;
        out->step = (size_t)out->cols * ch;
        out->mem = outputMemory;
        out->data = nullptr;
        if (dry)
            return SRX_OK;
        const size_t bytes = (size_t)out->rows * out->step;
        void* dev = results[slot(outPos)].get();
        if (outputMemory == SRX_MEM_DEVICE)
        {
            out->data = dev;
            SRX_HIP(hipStreamSynchronize(stream));
            return SRX_OK;
        }
        if (int rc = dst_host.reserve(bytes))
            return rc;
        SRX_HIP(hipMemcpyAsync(dst_host.get(), dev, bytes, hipMemcpyDeviceToHost, stream));
        SRX_HIP(hipStreamSynchronize(stream));
        out->data = dst_host.get();
    }
    return SRX_OK;
}

// ===========================================================================
// Independent TV-L1 jobs 0 .. n-1 on up to `in_flight` lanes side by side: job i runs as fn(flow, stream, i) on lane
// i % lanes, the first lane being the handle's own stream and workspace, every further lane on a host thread of its
// own (a job's launches are queued in portions with a look at the device's verdict in between, and the lanes must
// not wait for each other's looks).  Returns when every lane's stream is idle; the first failing lane's code and
// message are the call's.
template <typename Fn> static int run_flow_lanes(srx_handle* h, int n, int in_flight, Fn fn)
{
    const int lanes = std::max(1, std::min(std::min(in_flight, SRX_MAX_FLOW_PAIRS_IN_FLIGHT), n));
    while ((int)h->flow_lanes.size() < lanes - 1)
    {
        std::unique_ptr<srx_handle::FlowLane> lane(new srx_handle::FlowLane);
        SRX_HIP(hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking));
        h->flow_lanes.push_back(std::move(lane));
    }
    std::vector<int> rcs((size_t)lanes, SRX_OK);
    std::vector<std::string> errors((size_t)lanes);
    auto run_lane = [&](int lane) {
        Tvl1& flow = lane == 0 ? h->tvl1 : h->flow_lanes[(size_t)lane - 1]->tvl1;
        const hipStream_t st = lane == 0 ? h->stream : h->flow_lanes[(size_t)lane - 1]->stream;
        int rc = SRX_OK;
        if (lane > 0 && hipSetDevice(h->device) != hipSuccess)
        {
            set_error("flow lanes: hipSetDevice failed in a lane's thread");
            rc = SRX_ERR_DEVICE;
        }
        for (int i = lane; i < n && rc == SRX_OK; i += lanes)
            rc = fn(flow, st, i);
        if (hipStreamSynchronize(st) != hipSuccess && rc == SRX_OK)
        {
            set_error("flow lanes: stream synchronisation failed");
            rc = SRX_ERR_DEVICE;
        }
        rcs[(size_t)lane] = rc;
        if (rc != SRX_OK)
            errors[(size_t)lane] = last_error(); // the message is the thread's own
    };
    std::vector<std::thread> threads;
    for (int lane = 1; lane < lanes; ++lane)
        threads.emplace_back(run_lane, lane);
    if (n > 0)
        run_lane(0);
    for (std::thread& t : threads)
        t.join();
    for (int lane = 0; lane < lanes; ++lane)
        if (rcs[(size_t)lane] != SRX_OK)
        {
            set_error(errors[(size_t)lane]);
            return rcs[(size_t)lane];
        }
    return SRX_OK;
}

extern "C"
{

SRX_API const char* srx_last_error(void)
{
    return last_error().c_str();
}

SRX_API int srx_create_btvl1(int device, srx_handle** out)
{
    if (!out)
    {
        set_error("srx_create_btvl1: out is NULL");
        return SRX_ERR_INVALID;
    }
    *out = nullptr;
    std::unique_ptr<srx_handle> h(new srx_handle);
    h->device = device;
    h->dry = device < 0;
    if (!h->dry)
    {
        int n = 0;
        SRX_HIP(hipGetDeviceCount(&n));
        if (device >= n)
        {
            set_error("srx_create_btvl1: no such device");
            return SRX_ERR_INVALID;
        }
        SRX_HIP(hipSetDevice(device));
        SRX_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    }
    *out = h.release();
    return SRX_OK;
}

SRX_API void srx_destroy(srx_handle* h)
{
    if (!h)
        return;
    if (!h->dry)
    {
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
    }
    hipStream_t st = h->stream, st2 = h->stream2;
    hipEvent_t ev = h->prev_ready, ev2 = h->caller_read;
    if (st2)
        (void)hipStreamSynchronize(st2);
    std::vector<hipStream_t> lane_streams;
    for (auto& lane : h->flow_lanes)
    {
        (void)hipStreamSynchronize(lane->stream);
        lane_streams.push_back(lane->stream);
    }
    const bool dry = h->dry;
    delete h;
    for (hipStream_t s : lane_streams)
        (void)hipStreamDestroy(s);
    if (!dry && st)
        (void)hipStreamDestroy(st);
    if (st2)
        (void)hipStreamDestroy(st2);
    if (ev)
        (void)hipEventDestroy(ev);
    if (ev2)
        (void)hipEventDestroy(ev2);
}

// ---- parameters -----------------------------------------------------------
#define SRX_NEED(h)                                                                                                    \
    if (!(h))                                                                                                          \
    {                                                                                                                  \
        set_error("handle is NULL");                                                                                   \
        return SRX_ERR_INVALID;                                                                                        \
    }

static int* int_param(srx_handle* h, const std::string& n)
{
    if (n == "scale")
        return &h->p.scale;
    if (n == "iterations")
        return &h->p.iterations;
    if (n == "btvKernelSize")
        return &h->p.btvKernelSize;
    if (n == "blurKernelSize")
        return &h->p.blurKernelSize;
    if (n == "temporalAreaRadius")
        return &h->p.temporalAreaRadius;
    if (n == "outputMemory")
        return &h->outputMemory;
    if (n == "readAhead")
        return &h->readAhead;
    if (n == "flowPairsInFlight")
        return &h->flowPairsInFlight;
    return nullptr;
}
static double* double_param(srx_handle* h, const std::string& n)
{
    if (n == "tau")
        return &h->p.tau;
    if (n == "lambda")
        return &h->p.lambda;
    if (n == "alpha")
        return &h->p.alpha;
    if (n == "blurSigma")
        return &h->p.blurSigma;
    return nullptr;
}

SRX_API int srx_set_int(srx_handle* h, const char* name, int value)
{
    SRX_NEED(h);
    const std::string n = name ? name : "";
    if (n == "traceEvents")
    {
        h->trace = value != 0;
        return SRX_OK;
    }
    if (n == "kernelTiming")
    {
        h->solver.enable_timing(value != 0);
        h->solver.reset_timing();
        return SRX_OK;
    }
    if (int* p = int_param(h, n))
    {
        Params trial = h->p;
        const int keep = *p;
        *p = value;
        if (n == "outputMemory")
        {
            if (value != SRX_MEM_HOST && value != SRX_MEM_DEVICE)
            {
                *p = keep;
                set_error("outputMemory must be SRX_MEM_HOST or SRX_MEM_DEVICE");
                return SRX_ERR_INVALID;
            }
            return SRX_OK;
        }
        if (n == "readAhead")
        {
            *p = value != 0;
            return SRX_OK;
        }
        if (n == "flowPairsInFlight")
        {
            if (value < 1 || value > SRX_MAX_FLOW_PAIRS_IN_FLIGHT)
            {
                *p = keep;
                set_error("flowPairsInFlight must be 1 .. " + std::to_string(SRX_MAX_FLOW_PAIRS_IN_FLIGHT));
                return SRX_ERR_INVALID;
            }
            return SRX_OK;
        }
        if (int rc = validate_params(h->p))
        {
            h->p = trial;
            return rc;
        }
        return SRX_OK;
    }
    if (double_param(h, n)) // Algorithm::set(name, int) on a double parameter converts
        return srx_set_double(h, name, (double)value);
    set_error("unknown int parameter: " + n);
    return SRX_ERR_INVALID;
}

SRX_API int srx_set_double(srx_handle* h, const char* name, double value)
{
    SRX_NEED(h);
    const std::string n = name ? name : "";
    if (double* d = double_param(h, n))
    {
        const Params trial = h->p;
        *d = value;
        if (int rc = validate_params(h->p))
        {
            h->p = trial;
            return rc;
        }
        return SRX_OK;
    }
    set_error("unknown double parameter: " + n);
    return SRX_ERR_INVALID;
}

SRX_API int srx_get_int(srx_handle* h, const char* name, int* value)
{
    SRX_NEED(h);
    const std::string n = name ? name : "";
    if (!value)
    {
        set_error("value is NULL");
        return SRX_ERR_INVALID;
    }
    if (n == "traceEvents")
    {
        *value = h->trace;
        return SRX_OK;
    }
    if (n == "kernelTiming")
    {
        *value = h->solver.timing();
        return SRX_OK;
    }
    if (int* p = int_param(h, n))
    {
        *value = *p;
        return SRX_OK;
    }
    set_error("unknown int parameter: " + n);
    return SRX_ERR_INVALID;
}

SRX_API int srx_get_double(srx_handle* h, const char* name, double* value)
{
    SRX_NEED(h);
    const std::string n = name ? name : "";
    if (!value)
    {
        set_error("value is NULL");
        return SRX_ERR_INVALID;
    }
    if (double* d = double_param(h, n))
    {
        *value = *d;
        return SRX_OK;
    }
    set_error("unknown double parameter: " + n);
    return SRX_ERR_INVALID;
}

SRX_API int srx_set_flow_builtin(srx_handle* h, int kind)
{
    SRX_NEED(h);
    if (kind != SRX_FLOW_NONE && kind != SRX_FLOW_TVL1 && kind != SRX_FLOW_FARNEBACK)
    {
        set_error("unknown built-in flow");
        return SRX_ERR_INVALID;
    }
    h->flow_kind = kind;
    h->flow_fn = nullptr;
    h->flow2_fn = nullptr;
    h->flow_user = nullptr;
    return SRX_OK;
}

SRX_API int srx_set_flow_callback(srx_handle* h, srx_flow_fn fn, void* user)
{
    SRX_NEED(h);
    if (!fn)
    {
        set_error("flow callback is NULL");
        return SRX_ERR_INVALID;
    }
    h->flow_kind = SRX_FLOW_CALLBACK;
    h->flow_fn = fn;
    h->flow2_fn = nullptr;
    h->flow_user = user;
    return SRX_OK;
}

SRX_API int srx_set_flow_callback2(srx_handle* h, srx_flow2_fn fn, void* user)
{
    SRX_NEED(h);
    if (!fn)
    {
        set_error("flow callback is NULL");
        return SRX_ERR_INVALID;
    }
    h->flow_kind = SRX_FLOW_CALLBACK;
    h->flow_fn = nullptr;
    h->flow2_fn = fn;
    h->flow_user = user;
    return SRX_OK;
}

SRX_API int srx_flow_set_int(srx_handle* h, const char* name, int value)
{
    SRX_NEED(h);
    bool known = false;
    if (int rc = farneback_set_int(h->fb_params, name ? name : "", value, &known))
        return rc;
    if (known)
        return SRX_OK;
    return tvl1_set_int(h->flow_params, name ? name : "", value);
}
SRX_API int srx_flow_set_double(srx_handle* h, const char* name, double value)
{
    SRX_NEED(h);
    bool known = false;
    if (int rc = farneback_set_double(h->fb_params, name ? name : "", value, &known))
        return rc;
    if (known)
        return SRX_OK;
    return tvl1_set_double(h->flow_params, name ? name : "", value);
}
SRX_API int srx_flow_get_int(srx_handle* h, const char* name, int* value)
{
    SRX_NEED(h);
    if (value && farneback_get_int(h->fb_params, name ? name : "", value))
        return SRX_OK;
    return tvl1_get_int(h->flow_params, name ? name : "", value);
}
SRX_API int srx_flow_get_double(srx_handle* h, const char* name, double* value)
{
    SRX_NEED(h);
    if (value && farneback_get_double(h->fb_params, name ? name : "", value))
        return SRX_OK;
    return tvl1_get_double(h->flow_params, name ? name : "", value);
}

// ---- streaming --------------------------------------------------------------
SRX_API int srx_set_input(srx_handle* h, srx_next_frame_fn next, srx_reset_fn reset, void* user)
{
    SRX_NEED(h);
    h->next = next;
    h->reset = reset;
    h->user = user;
    h->firstCall = true;
    h->look_valid = h->look_end = false;
    return SRX_OK;
}

SRX_API int srx_reset(srx_handle* h)
{
    SRX_NEED(h);
    if (h->reset)
        h->reset(h->user);
    h->firstCall = true;
    h->look_valid = h->look_end = false;
    return SRX_OK;
}

SRX_API int srx_set_output_range(srx_handle* h, int first, int last)
{
    SRX_NEED(h);
    h->out_first = first;
    h->out_last = last;
    return SRX_OK;
}

// Caller memory handed over during a call (source frames, flow plug-in results) may be reused once the call
// returns: whatever reads of it are still queued on the handle's stream are waited for on every way out.
static int release_caller_memory(srx_handle* h, int rc)
{
    if (h->dry)
        return rc;
    if (rc != SRX_OK)
    {
        // an error return may leave copies of caller memory queued that no event marks
        if (hipStreamSynchronize(h->stream) != hipSuccess)
            (void)hipGetLastError();
        h->caller_read_pending = false;
        return rc;
    }
    return h->wait_caller_reads();
}

static int next_frame_impl(srx_handle* h, srx_image* out);
static int process_batch_impl(srx_handle* h, const srx_image* frames, int n, const srx_image* motions, int base_idx,
                              srx_image* out);

SRX_API int srx_next_frame(srx_handle* h, srx_image* out)
{
    SRX_NEED(h);
    return release_caller_memory(h, next_frame_impl(h, out));
}

static int next_frame_impl(srx_handle* h, srx_image* out)
{
    if (!out)
    {
        set_error("out is NULL");
        return SRX_ERR_INVALID;
    }
    std::memset(out, 0, sizeof(*out));
    if (!h->next)
    {
        // the reference starts with a NullFrameSource, whose empty frame trips the
        // warm-up assertion (lib/src/super_resolution.cpp:127, lib/src/btv_l1.cpp:479)
        set_error("no frame source set");
        return SRX_ERR_SOURCE;
    }
    if (int rc = h->set_device())
        return rc;
    if (h->firstCall)
    {
        if (int rc = h->init_impl())
            return rc;
        h->firstCall = false;
    }
    for (;;)
    {
        if (h->out_first <= h->out_last && h->outPos >= h->out_last)
            return SRX_OK; // span finished: empty frame
        bool produced = false;
        if (int rc = h->process_impl(out, &produced))
            return rc;
        if (!produced)
            return SRX_OK; // end of stream
        if (h->in_range(h->outPos))
            return SRX_OK;
    }
}

SRX_API int srx_drain_events(srx_handle* h, int* buf, int max_events)
{
    if (!h || !buf || max_events < 0)
        return 0;
    const int n = std::min<int>(max_events, (int)h->events.size());
    for (int i = 0; i < n; ++i)
        std::memcpy(buf + 5 * i, h->events[i].v, sizeof(int) * 5);
    h->events.erase(h->events.begin(), h->events.begin() + n);
    return n;
}

// ---- batch -------------------------------------------------------------------
SRX_API int srx_process(srx_handle* h, const srx_image* frames, int n, const srx_image* motions, int base_idx,
                        srx_image* out)
{
    SRX_NEED(h);
    return release_caller_memory(h, process_batch_impl(h, frames, n, motions, base_idx, out));
}

static int process_batch_impl(srx_handle* h, const srx_image* frames, int n, const srx_image* motions, int base_idx,
                              srx_image* out)
{
    // a dry-run handle checks the arguments and calls the flow plug-in as often as a real one, and computes nothing
    const bool dry = h->dry;
    if (!frames || !out || n < 1 || n > kMaxWindow || base_idx < 0 || base_idx >= n)
    {
        set_error("srx_process: bad frame list or base index");
        return SRX_ERR_INVALID;
    }
    if (int rc = h->set_device())
        return rc;
    if (int rc = validate_params(h->p))
        return rc;
    const srx_image& f0 = frames[0];
    for (int i = 0; i < n; ++i)
    {
        if (int rc = check_image(frames[i], "srx_process frame"))
            return rc;
        if (frames[i].rows != f0.rows || frames[i].cols != f0.cols || frames[i].channels != f0.channels ||
            frames[i].depth != f0.depth)
        {
            set_error("srx_process: frames differ in size or type");
            return SRX_ERR_INVALID;
        }
    }
    if (f0.channels != 1 && f0.channels != 3)
    {
        set_error("frames must have 1 or 3 channels");
        return SRX_ERR_INVALID;
    }
    const int rows = f0.rows, cols = f0.cols, ch = f0.channels;
    const size_t px = (size_t)rows * cols;
    if ((int)h->batch_frames.size() < n)
    {
        h->batch_frames.resize(n);
        h->batch_motions.resize(n);
        h->batch_frames8.resize(n);
    }
    const float* y[kMaxWindow];
    const float* m[kMaxWindow];
    const uint8_t* y8[kMaxWindow];
    for (int i = 0; i < n && !dry; ++i)
    {
        if (int rc = h->batch_frames[i].reserve(px * ch * sizeof(float)))
            return rc;
        if (int rc = h->to_float(frames[i], h->batch_frames[i].as<float>(), &h->batch_frames8[i]))
            return rc;
        y[i] = h->batch_frames[i].as<float>();
        y8[i] = h->batch_frames8[i].as<uint8_t>();
    }
    const int B = h->p.btvKernelSize, s = h->p.scale;
    const int orows = rows * s - 2 * B, ocols = cols * s - 2 * B;
    if (int rc = check_image(*out, "srx_process out"))
        return rc;
    if (out->rows != orows || out->cols != ocols || out->channels != ch || out->depth != f0.depth)
    {
        set_error("srx_process: out must be (rows*scale-2B) x (cols*scale-2B) with the frames' type");
        return SRX_ERR_INVALID;
    }
    int rc;
    if (motions)
    {
        for (int i = 0; i < n - 1; ++i)
        {
            if ((rc = check_image(motions[i], "srx_process motion")))
                return rc;
            if (motions[i].rows != rows || motions[i].cols != cols || motions[i].channels != 2 ||
                motions[i].depth != SRX_32F)
            {
                set_error("srx_process: motions must be rows x cols x 2 float");
                return SRX_ERR_INVALID;
            }
            if (dry)
                continue;
            if ((rc = h->batch_motions[i].reserve(px * 2 * sizeof(float))))
                return rc;
            if ((rc = copy_to_device(motions[i], h->batch_motions[i].get(), h->stream)))
                return rc;
            m[i] = h->batch_motions[i].as<float>();
        }
        if (dry)
            return SRX_OK;
        if (f0.depth == SRX_8U)
            h->solver.byte_frames(y8, n);
        rc = h->solver.process(y, m, n, base_idx, rows, cols, ch, h->p, nullptr, nullptr, h->stream);
    }
    else
    {
        // flow frame i -> base for every i != base, zero for the base (lib/src/btv_l1.cpp:311-321); the built-in
        // TV-L1 computes these independent pairs side by side (flowPairsInFlight of them)
        const bool side_by_side = !dry && h->flow_kind == SRX_FLOW_TVL1 && h->flowPairsInFlight > 1 && n > 2;
        for (int i = 0; i < n; ++i)
        {
            if (dry)
            {
                if (i != base_idx && (rc = h->calc_flow(frames[i], frames[base_idx], nullptr)))
                    return rc;
                continue;
            }
            if ((rc = h->batch_motions[i].reserve(px * 2 * sizeof(float))))
                return rc;
            if (i == base_idx)
                SRX_HIP(hipMemsetAsync(h->batch_motions[i].get(), 0, px * 2 * sizeof(float), h->stream));
            else if (!side_by_side && (rc = h->calc_flow(frames[i], frames[base_idx], h->batch_motions[i].as<float>())))
                return rc;
            m[i] = h->batch_motions[i].as<float>();
        }
        if (dry)
            return SRX_OK;
        if (side_by_side)
        {
            int others[kMaxWindow], n_others = 0;
            for (int i = 0; i < n; ++i)
                if (i != base_idx)
                    others[n_others++] = i;
            rc = run_flow_lanes(h, n_others, h->flowPairsInFlight, [&](Tvl1& flow, hipStream_t st, int j) {
                const int i = others[j];
                return flow.calc_images(frames[i], frames[base_idx], h->flow_params, h->batch_motions[i].as<float>(), nullptr, st);
            });
            if (rc)
                return rc;
        }
        if (f0.depth == SRX_8U)
            h->solver.byte_frames(y8, n);
        rc = h->solver.run(y, m, n, base_idx, rows, cols, ch, h->p, nullptr, nullptr, h->stream);
    }
    if (rc)
        return rc;
    const size_t obytes = (size_t)orows * ocols * ch * elem_size(f0.depth);
    if ((rc = h->batch_out.reserve(obytes)))
        return rc;
    if ((rc = launch_crop_convert(h->solver.last_x(), rows * s, cols * s, ch, B, h->batch_out.get(), f0.depth,
                                  h->solver.last_x_plane_stride(), h->stream)))
        return rc;
    if ((rc = copy_from_device(h->batch_out.get(), *out, h->stream)))
        return rc;
    SRX_HIP(hipStreamSynchronize(h->stream));
    return SRX_OK;
}

SRX_API int srx_flow_calc(srx_handle* h, const srx_image* frame0, const srx_image* frame1, srx_image* flow1,
                          srx_image* flow2)
{
    SRX_NEED(h);
    if (h->dry)
    {
        set_error("srx_flow_calc needs a GPU handle");
        return SRX_ERR_STATE;
    }
    if (!frame0 || !frame1 || !flow1)
    {
        set_error("srx_flow_calc: NULL argument");
        return SRX_ERR_INVALID;
    }
    if (int rc = h->set_device())
        return rc;
    if (int rc = check_image(*frame0, "flow frame0"))
        return rc;
    if (int rc = check_image(*frame1, "flow frame1"))
        return rc;
    if (frame0->rows != frame1->rows || frame0->cols != frame1->cols || frame0->channels != frame1->channels ||
        frame0->depth != frame1->depth)
    {
        set_error("srx_flow_calc: frames differ in size or type");
        return SRX_ERR_INVALID;
    }
    const int rows = frame0->rows, cols = frame0->cols;
    const size_t px = (size_t)rows * cols;
    if (int rc = h->flow_dev.reserve(px * 2 * sizeof(float)))
        return rc;
    if (int rc = h->calc_flow(*frame0, *frame1, h->flow_dev.as<float>()))
        return rc;
    if (int rc = check_image(*flow1, "flow1"))
        return rc;
    if (!flow2)
    {
        if (flow1->rows != rows || flow1->cols != cols || flow1->channels != 2 || flow1->depth != SRX_32F)
        {
            set_error("flow1 must be rows x cols x 2 float");
            return SRX_ERR_INVALID;
        }
        if (int rc = copy_from_device(h->flow_dev.get(), *flow1, h->stream))
            return rc;
    }
    else
    {
        // planar u, v (lib/src/optical_flow.cpp:77-80)
        if (int rc = check_image(*flow2, "flow2"))
            return rc;
        for (const srx_image* f : {flow1, flow2})
            if (f->rows != rows || f->cols != cols || f->channels != 1 || f->depth != SRX_32F)
            {
                set_error("flow1/flow2 must be rows x cols x 1 float");
                return SRX_ERR_INVALID;
            }
        if (int rc = h->staging.reserve(px * 2 * sizeof(float)))
            return rc;
        float* u = h->staging.as<float>();
        float* v = u + px;
        if (int rc = launch_split2(h->flow_dev.as<float>(), px, u, v, h->stream))
            return rc;
        if (int rc = copy_from_device(u, *flow1, h->stream))
            return rc;
        if (int rc = copy_from_device(v, *flow2, h->stream))
            return rc;
    }
    SRX_HIP(hipStreamSynchronize(h->stream));
    return SRX_OK;
}

SRX_API int srx_flow_calc_device(srx_handle* h, const float* gray0, const float* gray1, int rows, int cols,
                                 float* flow, long long* executed_iterations)
{
    SRX_NEED(h);
    if (h->dry || !gray0 || !gray1 || !flow)
    {
        set_error("srx_flow_calc_device: needs a GPU handle and non-NULL images");
        return SRX_ERR_INVALID;
    }
    if (int rc = h->set_device())
        return rc;
    if (h->flow_kind == SRX_FLOW_FARNEBACK)
    {
        if (int rc = h->farneback.calc_gray(gray0, gray1, rows, cols, h->fb_params, flow, h->stream))
            return rc;
        if (executed_iterations)
            *executed_iterations = 0;
    }
    else if (int rc = h->tvl1.calc_gray(gray0, gray1, rows, cols, h->flow_params, flow, executed_iterations, h->stream))
        return rc;
    h->pairs_pixel_iterations = -1;
    SRX_HIP(hipStreamSynchronize(h->stream));
    return SRX_OK;
}

SRX_API int srx_flow_calc_device_pairs(srx_handle* h, int n, const float* const* gray0, const float* const* gray1,
                                       int rows, int cols, float* const* flows, long long* executed_iterations,
                                       int in_flight)
{
    SRX_NEED(h);
    if (h->dry || n < 0 || (n > 0 && (!gray0 || !gray1 || !flows)))
    {
        set_error("srx_flow_calc_device_pairs: needs a GPU handle, n >= 0 and non-NULL lists");
        return SRX_ERR_INVALID;
    }
    if (in_flight < 1 || in_flight > SRX_MAX_FLOW_PAIRS_IN_FLIGHT)
    {
        set_error("srx_flow_calc_device_pairs: in_flight must be 1 .. " + std::to_string(SRX_MAX_FLOW_PAIRS_IN_FLIGHT));
        return SRX_ERR_INVALID;
    }
    for (int i = 0; i < n; ++i)
        if (!gray0[i] || !gray1[i] || !flows[i])
        {
            set_error("srx_flow_calc_device_pairs: pair " + std::to_string(i) + " has a NULL image");
            return SRX_ERR_INVALID;
        }
    if (int rc = h->set_device())
        return rc;
    if (h->flow_kind == SRX_FLOW_FARNEBACK) // one workspace: pair after pair
    {
        for (int i = 0; i < n; ++i)
        {
            if (int rc = h->farneback.calc_gray(gray0[i], gray1[i], rows, cols, h->fb_params, flows[i], h->stream))
                return rc;
            if (executed_iterations)
                executed_iterations[i] = 0;
        }
        h->pairs_pixel_iterations = 0;
        SRX_HIP(hipStreamSynchronize(h->stream));
        return SRX_OK;
    }
    std::vector<long long> work((size_t)std::max(n, 1), 0);
    const int rc = run_flow_lanes(h, n, in_flight, [&](Tvl1& flow, hipStream_t st, int i) {
        long long iters = 0;
        if (int rc = flow.calc_gray(gray0[i], gray1[i], rows, cols, h->flow_params, flows[i], &iters, st))
            return rc;
        work[(size_t)i] = flow.pixel_iterations();
        if (executed_iterations)
            executed_iterations[i] = iters;
        return (int)SRX_OK;
    });
    if (rc)
        return rc;
    h->pairs_pixel_iterations = 0;
    for (int i = 0; i < n; ++i)
        h->pairs_pixel_iterations += work[(size_t)i];
    return SRX_OK;
}

SRX_API int srx_flow_pixel_iterations(srx_handle* h, long long* pixel_iterations)
{
    SRX_NEED(h);
    if (!pixel_iterations)
    {
        set_error("srx_flow_pixel_iterations: NULL argument");
        return SRX_ERR_INVALID;
    }
    *pixel_iterations = h->pairs_pixel_iterations >= 0 ? h->pairs_pixel_iterations : h->tvl1.pixel_iterations();
    return SRX_OK;
}

SRX_API int srx_collect_garbage(srx_handle* h)
{
    SRX_NEED(h);
    if (h->dry)
        return SRX_OK;
    if (int rc = h->set_device())
        return rc;
    SRX_HIP(hipStreamSynchronize(h->stream));
    h->tvl1.release();
    for (auto& lane : h->flow_lanes)
        lane->tvl1.release();
    h->farneback.release();
    h->staging.release();
    h->flow_dev.release();
    h->batch_out.release();
    h->batch_frames.clear();
    h->batch_frames8.clear();
    h->batch_motions.clear();
    return SRX_OK;
}

SRX_API int srx_kernel_times(srx_handle* h, double* forward_ms, double* backproject_ms, long* iterations)
{
    SRX_NEED(h);
    if (h->dry)
    {
        set_error("srx_kernel_times needs a GPU handle");
        return SRX_ERR_STATE;
    }
    if (int rc = h->set_device())
        return rc;
    if (int rc = h->solver.collect_timing(h->stream))
        return rc;
    if (forward_ms)
        *forward_ms = h->solver.fwd_ms();
    if (backproject_ms)
        *backproject_ms = h->solver.bwd_ms();
    if (iterations)
        *iterations = h->solver.timed_iterations();
    return SRX_OK;
}

SRX_API int srx_k_copy(const void* src, void* dst, size_t bytes, void* stream)
{
    if (!src || !dst || bytes % 16 != 0 || ((uintptr_t)src | (uintptr_t)dst) % 16 != 0)
    {
        set_error("srx_k_copy: pointers and size must be multiples of 16 bytes");
        return SRX_ERR_INVALID;
    }
    return launch_copy16(src, dst, bytes, (hipStream_t)stream) ? SRX_ERR_DEVICE : SRX_OK;
}

SRX_API void* srx_stream(srx_handle* h)
{
    return h ? (void*)h->stream : nullptr;
}

SRX_API int srx_synchronize(srx_handle* h)
{
    SRX_NEED(h);
    if (h->dry)
        return SRX_OK;
    SRX_HIP(hipSetDevice(h->device));
    SRX_HIP(hipStreamSynchronize(h->stream));
    return SRX_OK;
}

// ---- pure host helpers ---------------------------------------------------------
SRX_API int srx_ring_slot(int index, int size)
{
    if (size <= 0)
        return 0;
    int m = index % size;
    return m < 0 ? m + size : m;
}

SRX_API void srx_window(int idx, int radius, int store_pos, int* start, int* end, int* base)
{
    const int s = std::max(idx - radius, 0);
    const int e = std::min(s + 2 * radius, store_pos);
    if (start)
        *start = s;
    if (end)
        *end = e;
    if (base)
        *base = idx - s;
}

SRX_API int srx_calc_btv_weights(int btv_kernel_size, double alpha, float* w)
{
    if (btv_kernel_size < 1 || !w)
        return 0;
    return calc_btv_weights(btv_kernel_size, alpha, w);
}

SRX_API int srx_gaussian_kernel(int n, double sigma, float* k)
{
    if (!k)
    {
        set_error("k is NULL");
        return SRX_ERR_INVALID;
    }
    return gaussian_kernel(n, sigma, k);
}

SRX_API int srx_span_plan(int total_frames, int radius, int world, int rank, int* out_first, int* out_last,
                          int* in_first, int* in_last)
{
    if (total_frames < 1 || radius < 0 || world < 1 || rank < 0 || rank >= world)
    {
        set_error("srx_span_plan: bad arguments");
        return SRX_ERR_INVALID;
    }
    // contiguous spans, the first (total % world) ranks get one frame more
    const int q = total_frames / world, r = total_frames % world;
    // Every rank feeds its handle the span plus its halos, and the handle's warm-up reads 2*radius+1 frames
    // (lib/src/btv_l1.cpp:476-481): a plan in which some rank gets fewer is refused for all ranks, so that
    // no rank of a job starts a stream its neighbour cannot start.  The ranks at the ends of the stream have one
    // halo only and therefore need radius+1 frames of their own, the others one.
    if (world > 1)
        for (int g = 0; g < world; ++g)
        {
            const int ga = g * q + std::min(g, r), gb = ga + q + (g < r ? 1 : 0) - 1;
            const int fed = std::min(std::max(gb, ga) + radius, total_frames - 1) - std::max(ga - radius, 0) + 1;
            if (gb < ga || fed < 2 * radius + 1)
            {
                set_error("srx_span_plan: spans too short for the temporal window (every rank needs 2*radius+1 frames "
                          "including its halos: radius+1 own frames at the ends of the stream)");
                return SRX_ERR_INVALID;
            }
        }
    const int a = rank * q + std::min(rank, r);
    const int b = a + q + (rank < r ? 1 : 0) - 1; // may be a-1 (empty span) when world > total
    if (out_first)
        *out_first = a;
    if (out_last)
        *out_last = b;
    if (in_first)
        *in_first = std::max(a - radius, 0);
    if (in_last)
        *in_last = std::min(std::max(b, a) + radius, total_frames - 1);
    return SRX_OK;
}

// ---- stage-level device entry points ---------------------------------------------
SRX_API int srx_k_relative_motions(const float* const* fwd_host, float* const* rel_host, int n, int base_idx, int rows,
                                   int cols, void* stream)
{
    if (!rel_host || n < 1 || n > kMaxWindow || base_idx < 0 || base_idx >= n || (n > 1 && !fwd_host))
    {
        set_error("srx_k_relative_motions: bad arguments");
        return SRX_ERR_INVALID;
    }
    PtrList f{};
    MutPtrList r{};
    for (int i = 0; i < n; ++i)
    {
        if (i < n - 1)
            f.p[i] = fwd_host[i];
        r.p[i] = rel_host[i];
    }
    return launch_relative_motions(f, r, n, base_idx, (size_t)rows * cols * 2, (hipStream_t)stream);
}

SRX_API int srx_k_build_maps(const float* rel, int rows, int cols, int scale, int16_t* fwd_map, int16_t* bwd_map,
                             void* stream)
{
    if (!rel || !fwd_map || !bwd_map || rows < 1 || cols < 1 || scale < 2 || (long)rows * scale > 32767 ||
        (long)cols * scale > 32767)
    {
        set_error("srx_k_build_maps: bad arguments");
        return SRX_ERR_INVALID;
    }
    CubicTablesDev t;
    if (int rc = t.ensure(rows, cols, rows * scale, cols * scale, (hipStream_t)stream))
        return rc;
    if (int rc = launch_build_maps(rel, rows, cols, scale, t.view(), fwd_map, bwd_map, nullptr, nullptr, (hipStream_t)stream))
        return rc;
    SRX_HIP(hipStreamSynchronize((hipStream_t)stream)); // the tables die with this call
    return SRX_OK;
}

SRX_API int srx_k_resize_cubic(const float* src, int srows, int scols, int ch, float* dst, int drows, int dcols,
                               void* stream)
{
    if (!src || !dst || srows < 1 || scols < 1 || drows < 1 || dcols < 1)
    {
        set_error("srx_k_resize_cubic: bad arguments");
        return SRX_ERR_INVALID;
    }
    CubicTablesDev t;
    if (int rc = t.ensure(srows, scols, drows, dcols, (hipStream_t)stream))
        return rc;
    if (int rc = launch_resize_cubic(src, srows, scols, ch, t.view(), dst, drows, dcols, 0, (hipStream_t)stream))
        return rc;
    SRX_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SRX_OK;
}

static int make_btv(int btv_kernel_size, const float* w_host, BtvW& bw)
{
    const int radius = (btv_kernel_size - 1) / 2;
    if (btv_kernel_size < 1 || radius > kMaxBtvRadius || !w_host)
    {
        set_error("btvKernelSize must be in 1..16 and weights non-NULL");
        return SRX_ERR_INVALID;
    }
    std::memset(&bw, 0, sizeof(bw));
    bw.radius = radius;
    int cnt = 0;
    for (int m = 0; m <= radius; ++m)
        cnt += radius + m + 1;
    std::memcpy(bw.w, w_host, sizeof(float) * cnt);
    return SRX_OK;
}

static int make_gauss(const float* gauss_host, int n, GaussK& g)
{
    if (!gauss_host || n < 1 || n > kMaxGauss || (n & 1) == 0)
    {
        set_error("gauss kernel must be odd, 1..31 taps");
        return SRX_ERR_INVALID;
    }
    std::memset(&g, 0, sizeof(g));
    g.n = n;
    std::memcpy(g.k, gauss_host, sizeof(float) * n);
    return SRX_OK;
}

SRX_API int srx_k_btv_regularization(const float* src, int rows, int cols, int ch, int btv_kernel_size,
                                     const float* w_host, float* dst, void* stream)
{
    if (!src || !dst || rows < 1 || cols < 1)
    {
        set_error("srx_k_btv_regularization: bad arguments");
        return SRX_ERR_INVALID;
    }
    BtvW bw;
    if (int rc = make_btv(btv_kernel_size, w_host, bw))
        return rc;
    return launch_btv_regularization(src, rows, cols, ch, bw, dst, (hipStream_t)stream);
}

SRX_API int srx_k_forward_sign(const float* X, int hr_rows, int hr_cols, int ch, const int16_t* bwd_map,
                               const float* y, int scale, const float* gauss_host, int gauss_n, uint32_t* sign_out,
                               void* stream)
{
    if (!X || !bwd_map || !y || !sign_out || scale < 2 || hr_rows < scale || hr_cols < scale ||
        hr_rows % scale || hr_cols % scale || hr_rows > 32767 || hr_cols > 32767 || (ch != 1 && ch != 3))
    {
        set_error("srx_k_forward_sign: bad arguments");
        return SRX_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    GaussK g;
    if (int rc = make_gauss(gauss_host, gauss_n, g))
        return rc;
    PtrList maps{}, ys{}, dmaps{};
    MutPtrList sg{};
    maps.p[0] = bwd_map;
    ys.p[0] = y;
    sg.p[0] = sign_out;
    BtvW bw7;
    float w7[kMaxBtvWeights] = {0};
    make_btv(7, w7, bw7);
    if (g_kernel_path == 0 && forward_fast_eligible(hr_rows, hr_cols, ch, scale, g) &&
        backproject_fast_eligible(hr_rows, hr_cols, ch, scale, g, bw7))
    {
        // the specialised kernels want a decoded map and X planar with a zero pixel behind every plane
        const size_t px = (size_t)hr_rows * hr_cols, ps = plane_stride_for(px);
        DeviceBuffer xpl, dec, pairs;
        if (int rc = xpl.reserve(ps * ch * sizeof(float)))
            return rc;
        if (int rc = dec.reserve(dec_bwd_words(px) * 4))
            return rc;
        if (int rc = pairs.reserve(sign_pair_bytes(hr_rows / 4, hr_cols / 4)))
            return rc;
        sg.p[0] = pairs.get();
        SRX_HIP(hipMemsetAsync(xpl.get(), 0, ps * ch * sizeof(float), st));
        if (int rc = launch_relayout(X, xpl.as<float>(), px, ch, ps, true, st))
            return rc;
        if (int rc = launch_decode_maps(nullptr, bwd_map, hr_rows, hr_cols, 4, nullptr, dec.as<uint32_t>(), st))
            return rc;
        dmaps.p[0] = dec.get();
        if (int rc = launch_forward_sign(xpl.as<float>(), hr_rows, hr_cols, ch, maps, &dmaps, ys, sg, 1, scale, g, 0, st))
            return rc;
        if (int rc = launch_sign_unpair(pairs.as<uint32_t>(), sign_out, hr_rows / 4, hr_cols / 4, st))
            return rc;
        SRX_HIP(hipStreamSynchronize(st)); // the temporaries die with this call
        return SRX_OK;
    }
    if (g_kernel_path == 0 && scale == 2 && g.n == 5)
    {
        // the scale-2 forward kernel reads X planar (with the zero pixel behind every plane)
        const size_t px = (size_t)hr_rows * hr_cols, ps = plane_stride_for(px);
        DeviceBuffer xpl;
        if (int rc = xpl.reserve(ps * ch * sizeof(float)))
            return rc;
        SRX_HIP(hipMemsetAsync(xpl.get(), 0, ps * ch * sizeof(float), st));
        if (int rc = launch_relayout(X, xpl.as<float>(), px, ch, ps, true, st))
            return rc;
        if (int rc = launch_forward_sign(xpl.as<float>(), hr_rows, hr_cols, ch, maps, nullptr, ys, sg, 1, scale, g, ps, st))
            return rc;
        SRX_HIP(hipStreamSynchronize(st));
        return SRX_OK;
    }
    return launch_forward_sign(X, hr_rows, hr_cols, ch, maps, nullptr, ys, sg, 1, scale, g, 0, st);
}

SRX_API int srx_k_backproject_update(const float* X, float* X_out, int hr_rows, int hr_cols, int ch,
                                     const int16_t* const* fwd_maps_host, const uint32_t* const* signs_host, int K,
                                     int scale, const float* gauss_host, int gauss_n, int btv_kernel_size,
                                     const float* w_host, double tau, double lambda, void* stream)
{
    if (!X || !X_out || X == X_out || !fwd_maps_host || !signs_host || K < 0 || K > kMaxWindow || scale < 2 ||
        hr_rows % scale || hr_cols % scale || hr_rows < scale || hr_cols < scale || hr_rows > 32767 ||
        hr_cols > 32767 || (ch != 1 && ch != 3))
    {
        set_error("srx_k_backproject_update: bad arguments");
        return SRX_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    GaussK g;
    if (int rc = make_gauss(gauss_host, gauss_n, g))
        return rc;
    BtvW bw;
    if (int rc = make_btv(btv_kernel_size, w_host, bw))
        return rc;
    PtrList maps{}, sg{}, dmaps{};
    for (int k = 0; k < K; ++k)
    {
        maps.p[k] = fwd_maps_host[k];
        sg.p[k] = signs_host[k];
    }
    if (g_kernel_path == 0 && backproject_fast_eligible(hr_rows, hr_cols, ch, scale, g, bw) &&
        (scale != 4 || forward_fast_eligible(hr_rows, hr_cols, ch, scale, g)))
    {
        // the specialised kernels want decoded maps, planar images, and the signs as pair planes (scale 4)
        // or as plain planes with a spare row (scale 2)
        const size_t px = (size_t)hr_rows * hr_cols, ps = plane_stride_for(px);
        const int lr_rows = hr_rows / scale, lr_cols = hr_cols / scale;
        const size_t padded = sign_pair_bytes(lr_rows, lr_cols);
        std::vector<DeviceBuffer> dec(K), spad(K);
        DeviceBuffer xin, xout;
        if (int rc = xin.reserve(ps * ch * sizeof(float)))
            return rc;
        if (int rc = xout.reserve(ps * ch * sizeof(float)))
            return rc;
        SRX_HIP(hipMemsetAsync(xin.get(), 0, ps * ch * sizeof(float), st));
        if (int rc = launch_relayout(X, xin.as<float>(), px, ch, ps, true, st))
            return rc;
        for (int k = 0; k < K; ++k)
        {
            if (int rc = dec[k].reserve(dec_fwd_words(hr_rows, hr_cols, scale) * 4))
                return rc;
            if (int rc = spad[k].reserve(padded))
                return rc;
            if (scale == 4)
            {
                if (int rc = launch_sign_pairs(signs_host[k], spad[k].as<uint32_t>(), lr_rows, lr_cols, st))
                    return rc;
            }
            else
                SRX_HIP(hipMemcpyAsync(spad[k].get(), signs_host[k], (size_t)lr_rows * lr_cols * 4,
                                       hipMemcpyDeviceToDevice, st));
            if (int rc = launch_decode_maps(fwd_maps_host[k], nullptr, hr_rows, hr_cols, scale, dec[k].as<uint32_t>(),
                                            nullptr, st))
                return rc;
            dmaps.p[k] = dec[k].get();
            sg.p[k] = spad[k].get();
        }
        if (int rc = launch_backproject_update(xin.as<float>(), xout.as<float>(), hr_rows, hr_cols, ch, maps, &dmaps,
                                               sg, K, scale, g, bw, tau, lambda, st))
            return rc;
        if (int rc = launch_relayout(xout.as<float>(), X_out, px, ch, ps, false, st))
            return rc;
        SRX_HIP(hipStreamSynchronize(st));
        return SRX_OK;
    }
    return launch_backproject_update(X, X_out, hr_rows, hr_cols, ch, maps, nullptr, sg, K, scale, g, bw, tau, lambda,
                                     st);
}

SRX_API int srx_quality_psnr(const srx_image* a, const srx_image* b, double* psnr, void* stream)
{
    if (!a || !b || !psnr)
    {
        set_error("srx_quality_psnr: NULL argument");
        return SRX_ERR_INVALID;
    }
    return quality_psnr(*a, *b, psnr, (hipStream_t)stream);
}

SRX_API int srx_quality_psnr_noalloc(const srx_image* a, const srx_image* b, double* psnr, void* stream)
{
    if (!a || !b || !psnr)
    {
        set_error("srx_quality_psnr_noalloc: NULL argument");
        return SRX_ERR_INVALID;
    }
    return quality_psnr_noalloc(*a, *b, psnr, (hipStream_t)stream);
}

SRX_API int srx_quality_mssim(const srx_image* a, const srx_image* b, double* per_channel, double* mssim, void* stream)
{
    if (!a || !b || (!per_channel && !mssim))
    {
        set_error("srx_quality_mssim: NULL argument");
        return SRX_ERR_INVALID;
    }
    return quality_mssim(*a, *b, per_channel, mssim, (hipStream_t)stream);
}

SRX_API int srx_set_kernel_path(int path)
{
    if (path != 0 && path != 1)
    {
        set_error("kernel path must be 0 (auto) or 1 (generic)");
        return SRX_ERR_INVALID;
    }
    g_kernel_path = path;
    return SRX_OK;
}

SRX_API int srx_get_kernel_path(void)
{
    return g_kernel_path;
}

SRX_API int srx_k_crop_convert(const float* X, int rows, int cols, int ch, int border, void* dst, int depth,
                               void* stream)
{
    if (!X || !dst || border < 0 || rows - 2 * border < 1 || cols - 2 * border < 1 ||
        (depth != SRX_8U && depth != SRX_16U && depth != SRX_32F))
    {
        set_error("srx_k_crop_convert: bad arguments");
        return SRX_ERR_INVALID;
    }
    return launch_crop_convert(X, rows, cols, ch, border, dst, depth, 0, (hipStream_t)stream);
}

SRX_API int srx_run(srx_handle* h, const float* const* y_host, const float* const* rel_host, int K, int base_idx,
                    int rows, int cols, int ch, float* out, float* x_full)
{
    SRX_NEED(h);
    if (h->dry || !y_host || !rel_host)
    {
        set_error("srx_run: needs a GPU handle and non-NULL pointer lists");
        return SRX_ERR_INVALID;
    }
    if (int rc = h->set_device())
        return rc;
    return h->solver.run(y_host, rel_host, K, base_idx, rows, cols, ch, h->p, out, x_full, h->stream);
}

SRX_API int srx_process_device(srx_handle* h, const float* const* y_host, const float* const* fwd_motions_host, int K,
                               int base_idx, int rows, int cols, int ch, float* out, float* x_full)
{
    SRX_NEED(h);
    if (h->dry || !y_host || (K > 1 && !fwd_motions_host))
    {
        set_error("srx_process_device: needs a GPU handle and non-NULL pointer lists");
        return SRX_ERR_INVALID;
    }
    if (int rc = h->set_device())
        return rc;
    return h->solver.process(y_host, fwd_motions_host, K, base_idx, rows, cols, ch, h->p, out, x_full, h->stream);
}

} // extern "C"
