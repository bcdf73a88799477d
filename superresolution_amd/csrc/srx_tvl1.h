// Built-in dense optical flow: duality-based TV-L1 (Zach, Pock, Bischof) on the GPU.
#pragma once

#include "srx_host.h"

namespace srx
{

struct Tvl1Params
{
    double tau = 0.25;
    double lambda = 0.15;
    double theta = 0.3;
    int nscales = 5;
    int warps = 5;
    double epsilon = 0.01;
    int iterations = 300;
    // inner iterations per kernel launch: 0 = by the size of the pyramid level (the default), 1, 4 or 8 = the
    // kernel of that many iterations at every level (same flows and iteration counts; for tests and tuning)
    int iterationsPerLaunch = 0;
};

int tvl1_set_int(Tvl1Params& p, const std::string& name, int value);
int tvl1_set_double(Tvl1Params& p, const std::string& name, double value);
int tvl1_get_int(const Tvl1Params& p, const std::string& name, int* value);
int tvl1_get_double(const Tvl1Params& p, const std::string& name, double* value);

class Tvl1
{
  public:
    // Flow frame0 -> frame1 for two images as a frame source delivers them
    // (host or device, 8U or 32F, 1/3/4 channels).  flow_uv: device, rows x
    // cols x 2 interleaved; executed_iters (optional, host) receives the
    // number of inner iterations actually run.
    int calc_images(const srx_image& f0, const srx_image& f1, const Tvl1Params& p, float* flow_uv,
                    long long* executed_iters, hipStream_t st);
    // Same on device-resident grey float images (rows x cols).
    int calc_gray(const float* I0, const float* I1, int rows, int cols, const Tvl1Params& p, float* flow_uv,
                  long long* executed_iters, hipStream_t st);
    // pixels x executed inner iterations summed over the levels and warps of the last call (the unit of the
    // flow stage's memory traffic: every inner iteration reads and writes its level's fields once)
    long long pixel_iterations() const;
    void release();

  private:
    struct Impl;
    std::shared_ptr<Impl> impl_;
};

// frame (device, dense rows) -> 8-bit grey as float, the way convertToType(frame, CV_8U, 1) reduces it
// (lib/src/input_array_utility.cpp:77-124); depth: SRX_8U / SRX_16U / SRX_32F
int launch_to_gray(const void* raw, size_t px, int ch, int depth, float* gray, hipStream_t st);

// interleaved (u,v) -> two planes
int launch_split2(const float* uv, size_t count, float* u, float* v, hipStream_t st);
// two planes -> interleaved (u,v)
int launch_merge2(const float* u, const float* v, size_t count, float* uv, hipStream_t st);
// dst[0 .. bytes) = src[0 .. bytes), bytes a multiple of 16, both 16-byte aligned (bandwidth reference)
int launch_copy16(const void* src, void* dst, size_t bytes, hipStream_t st);

} // namespace srx
