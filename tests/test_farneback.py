"""Built-in Farneback flow: oracle sanity (CPU) and GPU-vs-oracle parity.

The reference's default flow is a wrapper around OpenCV's calcOpticalFlowFarneback (lib/src/optical_flow.cpp:41-50,
84-87); OpenCV is not in the tree and the reference holds no expected flow fields, so parity with it is unpinned:
oracle/farneback_oracle.c restates the routine, the GPU kernels are held to it bit for bit, and the oracle itself is
checked against the known motion of synthetic pairs and against hand-derivable properties."""
import numpy as np
import pytest

from oracle import farneback as fb
from oracle import tvl1
from superresolution_amd import synth


def _pair(rows, cols, ch, seed):
    lr, flows, _ = synth.make_sequence(2, rows, cols, ch, 4, seed=seed)
    return lr[0].numpy(), lr[1].numpy(), flows[0].numpy()


def _wave(rows, cols, dx=0.0, dy=0.0):
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float64)
    x, y = xx - dx, yy - dy
    return np.round(127 + 60 * np.sin(x / 7.0) * np.cos(y / 9.0) + 40 * np.sin((x + y) / 5.0)).astype(np.float32)


def test_defaults_are_the_reference_class_defaults():
    p = fb.FarnebackParams()  # lib/src/optical_flow.cpp:41-50
    assert (p.pyrScale, p.numLevels, p.winSize, p.numIters, p.polyN, p.polySigma, p.flags) == (0.5, 5, 13, 10, 5, 1.1, 0)


def test_level_rule_and_smoothing_taps():
    L = fb._lib()
    # levels whose scaled width or height falls below 32 are dropped
    assert L.orc_farneback_levels(540, 960, 0.5, 5) == 4      # 33.75 rows at 1/16, 16.9 at 1/32
    assert L.orc_farneback_levels(64, 64, 0.5, 5) == 1
    assert L.orc_farneback_levels(63, 200, 0.5, 5) == 0
    assert L.orc_farneback_levels(1080, 1920, 0.5, 3) == 3
    k = fb.smooth_taps(1.0)
    assert list(k) == [0.25, 0.5, 0.25]                       # sigma 0, size 3: fixed table
    for scale, n in ((0.5, 3), (0.25, 9), (0.125, 19), (0.0625, 39)):  # round-half-even(5 sigma) | 1
        k = fb.smooth_taps(scale)
        assert len(k) == n and abs(float(k.sum()) - 1) < 1e-6 and np.array_equal(k, k[::-1])


def test_expansion_of_a_quadratic_is_exact():
    """A signal that IS a second-order polynomial is reproduced by its expansion: the coefficient planes are
    (d/dy, d/dx, yy, xx, xy) of it, away from the replicated border."""
    rows, cols = 40, 48
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float64)
    x, y = xx - 24, yy - 20
    img = (3 + 0.5 * x - 0.25 * y + 0.125 * x * x - 0.0625 * y * y + 0.03125 * x * y).astype(np.float32)
    R = fb.poly_exp(img, 5, 1.1)[8:-8, 8:-8]
    x, y = x[8:-8, 8:-8], y[8:-8, 8:-8]
    assert np.allclose(R[..., 0], -0.25 - 0.125 * y + 0.03125 * x, atol=2e-4)   # y
    assert np.allclose(R[..., 1], 0.5 + 0.25 * x + 0.03125 * y, atol=2e-4)      # x
    assert np.allclose(R[..., 2], -0.0625, atol=2e-4)                           # yy
    assert np.allclose(R[..., 3], 0.125, atol=2e-4)                             # xx
    assert np.allclose(R[..., 4], 0.03125, atol=2e-4)                           # xy


def test_inverse_moment_entries_invert_the_moment_matrix():
    g, xg, xxg, ig = fb.poly_setup(5, 1.1)
    full = np.concatenate([g[:0:-1], g]).astype(np.float64)
    xs = np.arange(-5, 6, dtype=np.float64)
    w = np.outer(full, full)
    X, Y = np.meshgrid(xs, xs)
    basis = [np.ones_like(X), X, Y, X * X, Y * Y, X * Y]
    G = np.array([[(w * a * b).sum() for b in basis] for a in basis])
    inv = np.linalg.inv(G)
    assert np.allclose([inv[1, 1], inv[0, 3], inv[3, 3], inv[5, 5]], ig, rtol=1e-6)


def test_update_flow_is_a_box_mean_and_a_2x2_solve():
    rows, cols, win = 20, 24, 5
    M = np.zeros((rows, cols, 5), np.float32)
    M[..., 0], M[..., 1], M[..., 2] = 2.0, 0.5, 3.0
    u, v = 0.75, -1.25   # flow (x, y); M holds G(y,y), G(x,y), G(x,x), h(y), h(x)
    M[..., 3] = 2.0 * v + 0.5 * u
    M[..., 4] = 0.5 * v + 3.0 * u
    fl = fb.update_flow(M, win)
    det = 2.0 * 3.0 - 0.25
    assert np.allclose(fl[..., 0], u * det / (det + 1e-3), atol=1e-6)
    assert np.allclose(fl[..., 1], v * det / (det + 1e-3), atol=1e-6)


def test_gaussian_window_weights_sum_to_one_and_solve_the_same_system():
    rows, cols, win = 20, 24, 7
    M = np.zeros((rows, cols, 5), np.float32)
    M[..., 0], M[..., 1], M[..., 2] = 2.0, 0.5, 3.0
    u, v = 0.75, -1.25
    M[..., 3] = 2.0 * v + 0.5 * u
    M[..., 4] = 0.5 * v + 3.0 * u
    fl = fb.update_flow(M, win, gaussian=True)
    det = 2.0 * 3.0 - 0.25
    assert np.allclose(fl[..., 0], u * det / (det + 1e-3), atol=1e-5)
    assert np.allclose(fl[..., 1], v * det / (det + 1e-3), atol=1e-5)
    # an impulse in h spreads as the outer product of the window's taps
    M[...] = 0
    M[..., 0] = M[..., 2] = 1.0
    M[10, 12, 4] = 1.0
    fl = fb.update_flow(M, win, gaussian=True)[..., 0]
    k = np.exp(-np.arange(-3, 4) ** 2 / (2 * 0.9 ** 2))
    k /= k.sum()
    assert np.allclose(fl[7:14, 9:16] * (1 + 1e-3), np.outer(k, k), atol=1e-6)


def test_oracle_recovers_a_translation_with_either_window():
    a, b = _wave(120, 160), _wave(120, 160, 1.5, -0.75)
    f = fb.calc_gray(a, b, fb.FarnebackParams(flags=256))[20:-20, 20:-20]
    assert abs(f[..., 0].mean() - 1.5) < 0.05 and abs(f[..., 1].mean() + 0.75) < 0.05


def test_oracle_recovers_a_translation():
    a, b = _wave(120, 160), _wave(120, 160, 1.5, -0.75)
    f = fb.calc_gray(a, b)[20:-20, 20:-20]
    assert abs(f[..., 0].mean() - 1.5) < 0.05 and abs(f[..., 1].mean() + 0.75) < 0.05
    assert f[..., 0].std() < 0.05 and f[..., 1].std() < 0.05


def test_oracle_recovers_known_motion_of_the_synthetic_clip():
    a, b, true = _pair(96, 128, 1, 3)
    est = fb.calc_gray(tvl1.to_gray(a), tvl1.to_gray(b))
    epe = np.sqrt(((est - true) ** 2).sum(-1))
    assert epe[12:-12, 12:-12].mean() < 0.15


def test_oracle_identical_frames_give_no_interior_flow():
    a = _wave(100, 120)
    f = fb.calc_gray(a, a)
    assert np.abs(f[30:-30, 30:-30]).max() < 1e-4


def test_bad_parameters_are_rejected():
    a = _wave(40, 40)
    for kw in (dict(pyrScale=1.0), dict(winSize=0), dict(polyN=0), dict(polyN=17), dict(flags=4)):
        with pytest.raises(ValueError):
            fb.calc_gray(a, a, fb.FarnebackParams(**kw))


def test_library_parameter_table_without_a_gpu():
    from superresolution_amd import api

    sr = api.BTV_L1(device=-1)
    sr.setOpticalFlow(api.Farneback(winSize=9, polySigma=1.5))
    iv, dv = api.C.c_int(), api.C.c_double()
    for name, want in (("numLevels", 5), ("winSize", 9), ("numIters", 10), ("polyN", 5), ("flags", 0)):
        api.L.check(sr._lib.srx_flow_get_int(sr._h, name.encode(), api.C.byref(iv)))
        assert iv.value == want
    for name, want in (("pyrScale", 0.5), ("polySigma", 1.5)):
        api.L.check(sr._lib.srx_flow_get_double(sr._h, name.encode(), api.C.byref(dv)))
        assert dv.value == want
    for name, bad in (("winSize", 0), ("polyN", 17), ("flags", 4)):
        with pytest.raises(Exception):
            sr.flowSet(name, bad)
    with pytest.raises(Exception):
        sr.flowSet("pyrScale", 1.0)
    # TV-L1 names still reach the TV-L1 table
    sr.flowSet("warps", 3)
    api.L.check(sr._lib.srx_flow_get_int(sr._h, b"warps", api.C.byref(iv)))
    assert iv.value == 3


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,ch,dtype,kw", [
    (64, 80, 1, np.uint8, {}),                                   # two levels
    (48, 72, 3, np.uint8, {}),                                   # one level: 24 rows at 1/2 is below 32
    (130, 150, 3, np.uint8, dict(numIters=3)),                   # three levels, odd sizes
    (97, 71, 1, np.float32, dict(winSize=7, polyN=7, polySigma=1.5, numIters=2)),
    (70, 90, 4, np.uint8, dict(pyrScale=0.7, numLevels=3, winSize=4, numIters=2)),
    (66, 68, 1, np.uint16, dict(numLevels=0, numIters=1)),
    (20, 12, 1, np.uint8, dict(numIters=2)),                     # smaller than the window and the border bands
    (80, 100, 3, np.uint8, dict(flags=256, numIters=4)),         # Gaussian window
    (50, 40, 1, np.uint8, dict(flags=256, winSize=21, numIters=2)),
])
def test_gpu_flow_matches_oracle_bit_exact(rows, cols, ch, dtype, kw):
    from superresolution_amd import api

    c3 = 3 if ch == 4 else ch
    a, b, _ = _pair(rows, cols, c3, 11 + rows)
    if ch == 4:
        a = np.concatenate([a, a[..., :1]], axis=-1)
        b = np.concatenate([b, b[..., :1]], axis=-1)
    if dtype == np.float32:
        a, b = a.astype(np.float32) / 255.0, b.astype(np.float32) / 255.0
    if dtype == np.uint16:
        a, b = a.astype(np.uint16) * 257, b.astype(np.uint16) * 257
    ref = fb.calc(a, b, fb.FarnebackParams(**kw))
    sr = api.BTV_L1()
    sr.setOpticalFlow(api.Farneback(**kw))
    got = sr.calcFlow(a, b)
    assert np.array_equal(ref, got)
    u, v = sr.calcFlow(a, b, planar=True)
    assert np.array_equal(ref[..., 0], u) and np.array_equal(ref[..., 1], v)


@pytest.mark.gpu
def test_gpu_stages_match_oracle_at_a_larger_size():
    """270x480 (four levels) through the device entry, and the recovered motion is the clip's."""
    import torch
    from superresolution_amd import api

    a, b, true = _pair(270, 480, 1, 5)
    ga, gb = tvl1.to_gray(a), tvl1.to_gray(b)
    ref = fb.calc_gray(ga, gb)
    sr = api.BTV_L1()
    sr.setOpticalFlow(api.Farneback())
    flow, n = sr.calcFlowDevice(torch.from_numpy(ga).cuda(), torch.from_numpy(gb).cuda())
    assert n == 0
    assert np.array_equal(flow.cpu().numpy(), ref)
    epe = np.sqrt(((ref - true) ** 2).sum(-1))
    assert epe[16:-16, 16:-16].mean() < 0.15


@pytest.mark.gpu
@pytest.mark.parametrize("read_ahead", [0, 1])
def test_streaming_with_farneback_flow_matches_oracle(read_ahead):
    """End to end: Farneback flow feeding the BTV-L1 solver, against the oracle driver with the oracle flow."""
    import oracle
    from oracle import stream as ostream
    from superresolution_amd import api

    n, r, T = 6, 1, 5
    lr, _, _ = synth.make_sequence(n, 40, 48, 3, 4, seed=21)
    lr = [f.numpy() for f in lr]
    kw = dict(numIters=3, winSize=9)
    p = fb.FarnebackParams(**kw)
    o = ostream.BTV_L1(lambda f0, f1: fb.calc(f0, f1, p))
    o.params = oracle.BtvParams(iterations=T)
    o.temporalAreaRadius = r
    o.setFrameSource(ostream.ListFrameSource(lr))
    sr = api.BTV_L1()
    sr.set("iterations", T)
    sr.set("temporalAreaRadius", r)
    sr.set("readAhead", read_ahead)
    sr.setOpticalFlow(api.Farneback(**kw))
    sr.setFrameSource(ostream.ListFrameSource(lr))
    for _ in range(n):
        a, b = o.nextFrame(), sr.nextFrame()
        assert np.array_equal(a, np.asarray(b))
    assert o.nextFrame() is None and sr.nextFrame() is None
