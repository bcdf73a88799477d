"""Quality metrics of the reference's quality harness (samples/quality_test/main.cpp:140-199, 320-324).

CPU part: the oracle against closed forms and an independent float64 evaluation (scipy).
GPU part: srx_quality_psnr / srx_quality_mssim through the C ABI against the oracle.
PSNR is exact (integer sums); MSSIM follows the reference's float operations, only the order of
the final double summation differs (<= 1e-9 for colour images; grey images allow 2e-7 because
cv::mean adds four floats at a time there).
Parity unpinned: the reference holds no expected metric values.
"""
import numpy as np
import pytest

import oracle


def _pair(rows, cols, ch, seed, noise=6):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:rows, 0:cols]
    base = 128 + 60 * np.sin(xx / 7.0 + seed) * np.cos(yy / 5.0) + 30 * np.sin((xx + yy) / 3.0)
    a = np.stack([np.clip(base + 10 * c, 0, 255) for c in range(ch)], axis=-1)
    b = np.clip(a + rng.normal(0, noise, a.shape), 0, 255)
    a8, b8 = a.astype(np.uint8), b.astype(np.uint8)
    if ch == 1:
        return a8[:, :, 0].copy(), b8[:, :, 0].copy()
    return a8, b8


def test_oracle_psnr_closed_form():
    a, b = _pair(33, 47, 3, 1)
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    assert oracle.psnr_8u(a, b) == pytest.approx(20 * np.log10(255 / (np.sqrt(mse) + np.finfo(np.float64).eps)),
                                                 rel=1e-15)
    # identical images: the DBL_EPSILON term keeps the value finite
    assert oracle.psnr_8u(a, a) == pytest.approx(20 * np.log10(255 / np.finfo(np.float64).eps), rel=1e-15)


def test_oracle_psnr_region_of_interest():
    a, b = _pair(40, 60, 3, 2)
    va, vb = a[7:-7, 7:-7], b[7:-7, 7:-7]
    assert oracle.psnr_8u(va, vb) == oracle.psnr_8u(np.ascontiguousarray(va), np.ascontiguousarray(vb))


def test_oracle_mssim_identical_images_and_the_reference_defect():
    a, _ = _pair(30, 41, 3, 3)
    v, m = oracle.mssim(a, a)
    assert np.all(m == 1.0)
    assert v == pytest.approx(2.0 / 3.0, rel=1e-15)   # (1 + 1 + mssim[3] = 0) / 3, main.cpp:198
    g, _ = _pair(30, 41, 1, 3)
    v, m = oracle.mssim(g, g)
    assert v == 1.0 and m[0] == 1.0


def test_oracle_mssim_against_float64_scipy():
    ndi = pytest.importorskip("scipy.ndimage")
    a, b = _pair(37, 52, 3, 4)
    x = np.arange(11) - 5
    k = np.exp(-x * x / (2 * 1.5 ** 2))
    k /= k.sum()

    def blur(im):
        t = ndi.correlate1d(im, k, axis=1, mode="mirror")
        return ndi.correlate1d(t, k, axis=0, mode="mirror")

    I1, I2 = a.astype(np.float64), b.astype(np.float64)
    mu1, mu2 = blur(I1), blur(I2)
    s11, s22, s12 = blur(I1 * I1) - mu1 * mu1, blur(I2 * I2) - mu2 * mu2, blur(I1 * I2) - mu1 * mu2
    ssim = ((2 * mu1 * mu2 + 6.5025) * (2 * s12 + 58.5225)) / ((mu1 * mu1 + mu2 * mu2 + 6.5025) * (s11 + s22 + 58.5225))
    ref = ssim.mean(axis=(0, 1))
    _, m = oracle.mssim(a, b)
    assert np.allclose(m, ref, atol=2e-5)


# ---------------------------------------------------------------------------
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("shape", [(64, 96, 3), (33, 47, 3), (70, 129, 1), (9, 13, 3), (540, 960, 3)])
def test_gpu_psnr_is_exact(shape):
    import torch
    from superresolution_amd import quality

    a, b = _pair(*shape, seed=5)
    want = oracle.psnr_8u(a, b)
    assert quality.PSNR(a, b) == want                                   # host images
    assert quality.PSNR(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()) == want
    assert quality.PSNR(a, a) == oracle.psnr_8u(a, a)


@gpu
def test_gpu_psnr_region_of_interest_without_copy():
    import torch
    from superresolution_amd import quality

    a, b = _pair(80, 120, 3, 6)
    B = 7
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    want = oracle.psnr_8u(a[B:-B, B:-B], b[B:-B, B:-B])
    assert quality.PSNR(ta[B:-B, B:-B], tb[B:-B, B:-B]) == want        # strided device views
    assert quality.PSNR(a[B:-B, B:-B], np.ascontiguousarray(b[B:-B, B:-B])) == want   # strided host view vs dense


@gpu
def test_gpu_psnr_rejects_float_images():
    from superresolution_amd import quality, _lib

    a = np.zeros((8, 8), np.float32)
    with pytest.raises(_lib.SrxError):
        quality.PSNR(a, a)


@gpu
@pytest.mark.parametrize("shape", [(64, 96, 3), (33, 47, 3), (70, 129, 1), (9, 13, 3), (4, 5, 1), (270, 480, 3)])
@pytest.mark.parametrize("as_float", [False, True])
def test_gpu_mssim_matches_oracle(shape, as_float):
    import torch
    from superresolution_amd import quality

    a, b = _pair(*shape, seed=7)
    want_ref, want_m = oracle.mssim(a, b)
    if as_float:
        a, b = a.astype(np.float32), b.astype(np.float32)
    tol = 2e-7 if shape[2] == 1 else 1e-9
    m, ref = quality.mssim_channels(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    assert np.allclose(m, want_m, rtol=0, atol=tol)
    assert abs(ref - want_ref) <= tol
    m2, ref2 = quality.mssim_channels(a, b)                             # host images: same partial sums
    assert np.array_equal(m2, m) and ref2 == ref
    assert quality.MSSIM(a, b) == ref
    assert quality.MSSIM(a, b, as_reference=False) == pytest.approx(float(np.mean(m)), rel=1e-15)


@gpu
def test_gpu_mssim_region_of_interest():
    import torch
    from superresolution_amd import quality

    a, b = _pair(90, 140, 3, 8)
    B = 7
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    _, want_m = oracle.mssim(a[B:-B, B:-B], b[B:-B, B:-B])
    m, _ = quality.mssim_channels(ta[B:-B, B:-B], tb[B:-B, B:-B])
    assert np.allclose(m, want_m, rtol=0, atol=1e-9)
