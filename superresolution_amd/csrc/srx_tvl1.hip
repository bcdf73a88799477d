// Built-in TV-L1 dense optical flow for gfx950.
#include "srx_tvl1.h"

namespace srx
{

namespace
{
__global__ void k_split2(const float2* __restrict__ uv, size_t count, float* __restrict__ u, float* __restrict__ v)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count)
        return;
    const float2 f = uv[e];
    u[e] = f.x;
    v[e] = f.y;
}

int bad_name(const std::string& n)
{
    set_error("unknown flow parameter: " + n);
    return SRX_ERR_INVALID;
}
} // namespace

int launch_split2(const float* uv, size_t count, float* u, float* v, hipStream_t st)
{
    const int bs = 256;
    hipLaunchKernelGGL(k_split2, dim3((unsigned)((count + bs - 1) / bs)), dim3(bs), 0, st, (const float2*)uv, count, u,
                       v);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "split2");
}

int tvl1_set_int(Tvl1Params& p, const std::string& n, int v)
{
    if (v < 1)
    {
        set_error("flow parameter " + n + " must be positive");
        return SRX_ERR_INVALID;
    }
    if (n == "nscales")
        p.nscales = v;
    else if (n == "warps")
        p.warps = v;
    else if (n == "iterations")
        p.iterations = v;
    else
        return bad_name(n);
    return SRX_OK;
}

int tvl1_set_double(Tvl1Params& p, const std::string& n, double v)
{
    if (!(v > 0.0))
    {
        set_error("flow parameter " + n + " must be positive");
        return SRX_ERR_INVALID;
    }
    if (n == "tau")
        p.tau = v;
    else if (n == "lambda")
        p.lambda = v;
    else if (n == "theta")
        p.theta = v;
    else if (n == "epsilon")
        p.epsilon = v;
    else
        return bad_name(n);
    return SRX_OK;
}

int tvl1_get_int(const Tvl1Params& p, const std::string& n, int* v)
{
    if (!v)
        return bad_name(n);
    if (n == "nscales")
        *v = p.nscales;
    else if (n == "warps")
        *v = p.warps;
    else if (n == "iterations")
        *v = p.iterations;
    else
        return bad_name(n);
    return SRX_OK;
}

int tvl1_get_double(const Tvl1Params& p, const std::string& n, double* v)
{
    if (!v)
        return bad_name(n);
    if (n == "tau")
        *v = p.tau;
    else if (n == "lambda")
        *v = p.lambda;
    else if (n == "theta")
        *v = p.theta;
    else if (n == "epsilon")
        *v = p.epsilon;
    else
        return bad_name(n);
    return SRX_OK;
}

struct Tvl1::Impl
{
};

int Tvl1::calc_images(const srx_image&, const srx_image&, const Tvl1Params&, float*, long long*, hipStream_t)
{
    set_error("built-in TV-L1 flow is not available in this build");
    return SRX_ERR_FLOW;
}

int Tvl1::calc_gray(const float*, const float*, int, int, const Tvl1Params&, float*, long long*, hipStream_t)
{
    set_error("built-in TV-L1 flow is not available in this build");
    return SRX_ERR_FLOW;
}

void Tvl1::release()
{
    impl_.reset();
}

} // namespace srx
