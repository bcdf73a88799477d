// Built-in dense optical flow: Farneback's polynomial-expansion method on the GPU, with the
// parameters of the reference's default flow class (lib/src/optical_flow.cpp:41-50,84-87).
#pragma once

#include "srx_host.h"

namespace srx
{

struct FarnebackParams
{
    double pyrScale = 0.5;
    int numLevels = 5;
    int winSize = 13;
    int numIters = 10;
    int polyN = 5;
    double polySigma = 1.1;
    int flags = 0;
};

// SRX_OK, or SRX_ERR_INVALID with the error text set; *known = false when the name is not a Farneback parameter
int farneback_set_int(FarnebackParams& p, const std::string& name, int value, bool* known);
int farneback_set_double(FarnebackParams& p, const std::string& name, double value, bool* known);
bool farneback_get_int(const FarnebackParams& p, const std::string& name, int* value);
bool farneback_get_double(const FarnebackParams& p, const std::string& name, double* value);

class Farneback
{
  public:
    // Flow frame0 -> frame1 for two images as a frame source delivers them (host or device,
    // 8U / 16U / 32F, 1/3/4 channels).  flow_uv: device, rows x cols x 2 interleaved.
    int calc_images(const srx_image& f0, const srx_image& f1, const FarnebackParams& p, float* flow_uv,
                    hipStream_t st);
    // Same on device-resident grey float images (rows x cols, values of an 8-bit image).
    int calc_gray(const float* I0, const float* I1, int rows, int cols, const FarnebackParams& p, float* flow_uv,
                  hipStream_t st);
    void release();

  private:
    struct Impl;
    std::shared_ptr<Impl> impl_;
};

} // namespace srx
