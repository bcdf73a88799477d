"""Stage-by-stage parity of the HIP kernels (through the C ABI) against the CPU oracle.

The oracle is a restatement of lib/src/btv_l1.cpp and the OpenCV routines it
calls; its own parity with the reference binary is unpinned (no OpenCV here),
so these tests pin GPU == oracle, bit for bit, on the same seeded inputs.
"""
import numpy as np
import pytest

import oracle
from conftest import smooth_motion

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[0, 1], ids=["auto", "generic"])
def kernel_path(request):
    """Every test runs twice: with the specialised kernels enabled (they are picked where the parameters allow)
    and with the generic kernels forced."""
    from superresolution_amd import ops as _ops

    _ops.set_kernel_path(request.param)
    yield request.param
    _ops.set_kernel_path(0)


@pytest.fixture(scope="module")
def ops():
    import torch  # noqa: F401
    from superresolution_amd import ops as _ops

    return _ops


def dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def test_host_helpers_match_oracle(ops):
    for B, alpha in [(7, 0.7), (5, 0.5), (3, 0.9), (1, 0.7), (9, 0.65), (15, 0.7), (4, 0.7)]:
        assert np.array_equal(ops.calc_btv_weights(B, alpha), oracle.calc_btv_weights(B, alpha))
    for n, sigma in [(1, 0), (3, 0), (5, 0), (7, 0), (9, 0), (5, 1.2), (7, 0.8), (11, 2.5), (3, 0.4)]:
        assert np.array_equal(ops.gaussian_kernel(n, sigma), oracle.gaussian_kernel(n, sigma))


@pytest.mark.parametrize("n,base", [(5, 2), (9, 4), (9, 0), (7, 6), (1, 0), (2, 1)])
def test_relative_motions(ops, rng, n, base):
    rows, cols = 19, 23
    fwd = [rng.standard_normal((rows, cols, 2)).astype(np.float32) for _ in range(n - 1)]
    ref = oracle.calc_relative_motions(fwd, base, rows, cols)
    got = ops.relative_motions([dev(f) for f in fwd], base, rows, cols)
    for r, g in zip(ref, got):
        assert np.array_equal(r, host(g))


@pytest.mark.parametrize("rows,cols,scale", [(24, 32, 4), (17, 29, 4), (30, 22, 2), (11, 13, 3), (5, 4, 4), (2, 3, 2)])
def test_build_maps_bit_exact(ops, rng, rows, cols, scale):
    rel = smooth_motion(rng, rows, cols, amp=2.0)
    # exact ties and near-ties for the round-half-even rule
    rel[0, 0] = (0.125, -0.125)  # 4 * 0.125 = 0.5 -> x +- 0.5
    rel[-1, -1] = (0.375, 0.625)
    hr = oracle.upscale_motion(rel, scale)
    f, b = oracle.build_motion_maps(hr)
    ref_f, ref_b = oracle.map_to_short(f), oracle.map_to_short(b)
    got_f, got_b = ops.build_maps(dev(rel), scale)
    assert np.array_equal(ref_f, host(got_f))
    assert np.array_equal(ref_b, host(got_b))


def test_build_maps_nan_and_huge(ops, rng):
    rows, cols, scale = 12, 16, 4
    rel = smooth_motion(rng, rows, cols)
    rel[3, 4] = (np.nan, 1e30)
    rel[7, 9] = (-1e30, np.inf)
    rel[10, 2] = (5000.0, -5000.0)  # saturates to +-32767/8 after the short cast
    hr = oracle.upscale_motion(rel, scale)
    f, b = oracle.build_motion_maps(hr)
    got_f, got_b = ops.build_maps(dev(rel), scale)
    assert np.array_equal(oracle.map_to_short(f), host(got_f))
    assert np.array_equal(oracle.map_to_short(b), host(got_b))


@pytest.mark.parametrize("rows,cols,ch,scale", [(24, 32, 1, 4), (17, 29, 3, 4), (30, 22, 3, 2), (11, 13, 1, 3),
                                                (3, 2, 3, 4), (1, 1, 1, 2)])
def test_resize_cubic(ops, rng, rows, cols, ch, scale):
    src = (rng.random((rows, cols, ch)) * 255).astype(np.float32)
    ref = oracle.resize_cubic(src, rows * scale, cols * scale)
    got = ops.resize_cubic(dev(src), rows * scale, cols * scale)
    assert np.array_equal(ref, host(got))


@pytest.mark.parametrize("rows,cols,ch,B,alpha", [(40, 56, 1, 7, 0.7), (33, 47, 3, 7, 0.7), (20, 20, 3, 5, 0.5),
                                                  (25, 31, 1, 3, 0.9), (16, 16, 1, 1, 0.7), (30, 30, 3, 9, 0.6),
                                                  (6, 40, 1, 7, 0.7), (7, 7, 3, 7, 0.7)])
def test_btv_regularization(ops, rng, rows, cols, ch, B, alpha):
    # quantised values so that equal neighbours (sign 0) actually occur
    src = np.round(rng.random((rows, cols, ch)) * 12).astype(np.float32) * 0.37
    w = oracle.calc_btv_weights(B, alpha)
    ref = oracle.btv_regularization(src, B, w)
    got = ops.btv_regularization(dev(src), B, w)
    assert np.array_equal(ref, host(got))


def _forward_ref(X, bwd_float, y, scale, gk):
    a = oracle.remap_nearest(X, bwd_float)
    b = oracle.sep_filter(a, gk)
    c = oracle.resize_nearest(b, y.shape[0], y.shape[1])
    return oracle.diff_sign(y, c)


@pytest.mark.parametrize("rows,cols,ch,scale,n,sigma", [(24, 32, 1, 4, 5, 0.0), (17, 21, 3, 4, 5, 0.0),
                                                        (20, 18, 3, 2, 5, 0.0), (12, 14, 1, 3, 3, 0.0),
                                                        (16, 16, 3, 4, 7, 0.0), (16, 12, 1, 4, 5, 1.3),
                                                        (10, 10, 3, 2, 9, 2.0), (8, 8, 1, 4, 1, 0.0),
                                                        # the scale-2 forward kernel: strip boundary at 62 LR
                                                        # columns, minimum size, a non-dyadic blur
                                                        (9, 62, 3, 2, 5, 0.0), (13, 63, 1, 2, 5, 0.0),
                                                        (4, 4, 3, 2, 5, 0.0), (11, 130, 3, 2, 5, 1.1)])
def test_forward_sign(ops, rng, rows, cols, ch, scale, n, sigma):
    H, W = rows * scale, cols * scale
    X = (rng.random((H, W, ch)) * 255).astype(np.float32)
    rel = smooth_motion(rng, rows, cols, amp=3.0)
    rel[0, :3] = -4.0  # push some gathers out of the image (constant border)
    hr = oracle.upscale_motion(rel, scale)
    _, bwd = oracle.build_motion_maps(hr)
    gk = oracle.gaussian_kernel(n, sigma)
    # y close to the projection so that all three sign values occur
    proj = oracle.resize_nearest(oracle.sep_filter(oracle.remap_nearest(X, bwd), gk), rows, cols)
    y = proj + rng.integers(-1, 2, proj.shape).astype(np.float32)
    ref = _forward_ref(X, bwd, y, scale, gk)
    assert set(np.unique(ref)) == {-1.0, 0.0, 1.0}
    _, bwd_map = ops.build_maps(dev(rel), scale)
    got = host(ops.forward_sign(dev(X), bwd_map, dev(y), scale, gk))
    assert np.array_equal(ref.astype(np.int8).reshape(rows, cols, ch), ops.unpack_signs(got, ch))
    assert np.array_equal(got, ops.pack_signs(ref.reshape(rows, cols, ch)))


@pytest.mark.parametrize("rows,cols,ch,kind", [(40, 70, 3, "large"), (40, 70, 1, "large"), (36, 130, 3, "jitter"),
                                               (24, 64, 1, "jitter"), (33, 65, 3, "outside"), (20, 127, 3, "wild"),
                                               (1, 1, 1, "small"), (2, 3, 3, "small"), (3, 63, 1, "small"),
                                               (5, 64, 3, "small"), (7, 127, 3, "small"), (4, 126, 1, "small")])
def test_forward_sign_beyond_the_lds_window(ops, rng, rows, cols, ch, kind):
    """Motions the specialised forward kernel cannot serve from its LDS window (sources more than 15 rows / 16
    columns away, neighbouring pixels whose sources are far apart, sources outside every border): the values
    must not depend on which path fetched them."""
    scale = 4
    H, W = rows * scale, cols * scale
    X = (rng.random((H, W, ch)) * 255).astype(np.float32)
    if kind == "small":      # sizes around the kernel's strip of 63 LR columns, down to one pixel
        rel = smooth_motion(rng, rows, cols, amp=0.7)
    elif kind == "large":    # smooth, up to ~40 HR pixels
        rel = smooth_motion(rng, rows, cols, amp=5.0)
    elif kind == "jitter":   # neighbouring LR pixels move independently: groups that are not runs, far groups
        rel = smooth_motion(rng, rows, cols, amp=1.0) + rng.standard_normal((rows, cols, 2)).astype(np.float32) * 3.0
    elif kind == "outside":  # the whole frame shifted so that bands at two borders leave the image
        rel = np.zeros((rows, cols, 2), np.float32)
        rel[..., 0], rel[..., 1] = 3.3, -2.6
    else:                    # int16 saturation of the map and everything in between
        rel = rng.standard_normal((rows, cols, 2)).astype(np.float32) * 40.0
        rel[3, 5] = (20000.0, -20000.0)
    hr = oracle.upscale_motion(rel, scale)
    _, bwd = oracle.build_motion_maps(hr)
    gk = oracle.gaussian_kernel(5, 0.0)
    proj = oracle.resize_nearest(oracle.sep_filter(oracle.remap_nearest(X, bwd), gk), rows, cols)
    y = proj + rng.integers(-1, 2, proj.shape).astype(np.float32)
    ref = _forward_ref(X, bwd, y, scale, gk)
    _, bwd_map = ops.build_maps(dev(rel), scale)
    got = host(ops.forward_sign(dev(X), bwd_map, dev(y), scale, gk))
    assert np.array_equal(got, ops.pack_signs(ref.reshape(rows, cols, ch)))


def _backproject_ref(X, signs, fwd_floats, scale, gk, B, w, tau, lam):
    diff = np.zeros_like(X)
    for s, f in zip(signs, fwd_floats):
        a = oracle.upscale(s, scale)
        b = oracle.sep_filter(a, gk)
        a = oracle.remap_nearest(b, f)
        diff = diff + a
    if lam > 0:
        reg = oracle.btv_regularization(X, B, w)
        diff = oracle.add_weighted(diff, 1.0, reg, -lam)
    return oracle.add_weighted(X, 1.0, diff, tau)


@pytest.mark.parametrize("rows,cols,ch,scale,n,sigma,K,lam", [(24, 32, 1, 4, 5, 0.0, 3, 0.03),
                                                              (17, 21, 3, 4, 5, 0.0, 5, 0.03),
                                                              (20, 18, 3, 2, 5, 0.0, 2, 0.03),
                                                              (12, 14, 1, 3, 3, 0.0, 3, 0.1),
                                                              (16, 16, 3, 4, 7, 0.0, 2, 0.03),
                                                              (16, 12, 1, 4, 5, 1.3, 3, 0.0),
                                                              (10, 10, 3, 2, 9, 2.0, 2, 0.03),
                                                              (9, 9, 1, 2, 5, 0.7, 9, 0.03),
                                                              # scale 2 with the dyadic 5-tap blur: the update
                                                              # kernel's 3x3-cell variant, down to its minimum size
                                                              (14, 22, 1, 2, 5, 0.0, 17, 0.03),
                                                              (4, 4, 3, 2, 5, 0.0, 3, 0.03),
                                                              (16, 34, 3, 2, 5, 0.0, 4, 0.0),
                                                              (6, 66, 3, 2, 5, 0.0, 2, 0.03)])
def test_backproject_update(ops, rng, rows, cols, ch, scale, n, sigma, K, lam):
    H, W = rows * scale, cols * scale
    B, alpha, tau = 7, 0.7, 1.3
    X = np.round(rng.random((H, W, ch)) * 255).astype(np.float32)
    gk = oracle.gaussian_kernel(n, sigma)
    w = oracle.calc_btv_weights(B, alpha)
    signs, fwd_floats, fwd_maps, signs_dev = [], [], [], []
    for k in range(K):
        rel = smooth_motion(rng, rows, cols, amp=3.0)
        if k == 0:
            rel[:2, :] = 6.0  # forward map leaves the image at the top/left
        hr = oracle.upscale_motion(rel, scale)
        f, _ = oracle.build_motion_maps(hr)
        s = rng.integers(-1, 2, (rows, cols, ch)).astype(np.float32)
        signs.append(s)
        fwd_floats.append(f)
        fm, _ = ops.build_maps(dev(rel), scale)
        fwd_maps.append(fm)
        signs_dev.append(dev(ops.pack_signs(s)))
    ref = _backproject_ref(X, signs, fwd_floats, scale, gk, B, w, tau, lam)
    got = host(ops.backproject_update(dev(X), fwd_maps, signs_dev, scale, gk, B, w, tau, lam))
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("ch,lam", [(3, 0.03), (1, 0.0)])
def test_backproject_update_scale_2_patches_that_fit_and_patches_that_do_not(ops, rng, ch, lam):
    """Scale 2, several tiles of the update kernel per axis.  Per tile and frame the kernel stages the patch of
    the sign plane that the tile's pixels reach in LDS when the patch is small enough, and gathers from global
    memory otherwise; the frames here make tiles of both kinds, next to each other and in one window: smooth
    motion, wild motion, smooth on the left and jittered on the right, and motion that leaves the image."""
    rows, cols, scale, K = 40, 80, 2, 5
    H, W = rows * scale, cols * scale
    B, alpha, tau = 7, 0.7, 1.3
    X = np.round(rng.random((H, W, ch)) * 255).astype(np.float32)
    gk = oracle.gaussian_kernel(5, 0.0)
    w = oracle.calc_btv_weights(B, alpha)
    rels = [smooth_motion(rng, rows, cols, amp=2.5),
            rng.standard_normal((rows, cols, 2)).astype(np.float32) * 12.0,
            smooth_motion(rng, rows, cols, amp=1.5),
            smooth_motion(rng, rows, cols, amp=1.0) + np.float32(9.0),
            np.zeros((rows, cols, 2), np.float32)]
    rels[2][:, cols // 2:] += rng.standard_normal((rows, cols - cols // 2, 2)).astype(np.float32) * 2.0
    rels[3][:, :, 1] -= np.float32(14.0)
    signs, fwd_floats, fwd_maps, signs_dev = [], [], [], []
    for rel in rels[:K]:
        hr = oracle.upscale_motion(rel, scale)
        f, _ = oracle.build_motion_maps(hr)
        s = rng.integers(-1, 2, (rows, cols, ch)).astype(np.float32)
        signs.append(s)
        fwd_floats.append(f)
        fm, _ = ops.build_maps(dev(rel), scale)
        fwd_maps.append(fm)
        signs_dev.append(dev(ops.pack_signs(s)))
    ref = _backproject_ref(X, signs, fwd_floats, scale, gk, B, w, tau, lam)
    got = host(ops.backproject_update(dev(X), fwd_maps, signs_dev, scale, gk, B, w, tau, lam))
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("ch,scale", [(1, 4), (3, 4), (3, 2)])
def test_backproject_update_with_values_that_are_not_plain(ops, rng, ch, scale):
    """NaN, infinite, huge, denormal and signed-zero pixels: diffSign of the reference compares, so a NaN
    counts as equal (lib/src/btv_l1.cpp:138-141); the specialised kernel has to give the same regulariser as
    the comparison-based oracle wherever the output is a number, and NaN where the oracle has NaN."""
    rows, cols, K = 12, 40, 3
    H, W = rows * scale, cols * scale
    B, alpha, tau, lam = 7, 0.7, 1.3, 0.03
    X = np.round(rng.random((H, W, ch)) * 255).astype(np.float32)
    X[:, : W // 2] = np.round(X[:, : W // 2] / 64) * 64  # many ties
    odd = [np.nan, np.inf, -np.inf, 3e38, -3e38, 1e30, -1e19, 1e-40, 2e-40, -1e-40, 1e-30, -0.0, 0.0, 1.5e-45]
    for i, v in enumerate(odd):
        X[8 + (i * 5) % (H - 16), 9 + (i * 11) % (W - 18), i % ch] = v
    X[H // 2, W // 2 : W // 2 + 6, :] = 1e-40  # a run of equal denormals next to zeros
    X[H // 2 + 1, W // 2 : W // 2 + 6, :] = 0.0
    gk = oracle.gaussian_kernel(5, 0.0)
    w = oracle.calc_btv_weights(B, alpha)
    signs, fwd_floats, fwd_maps, signs_dev = [], [], [], []
    for k in range(K):
        rel = smooth_motion(rng, rows, cols, amp=2.0)
        hr = oracle.upscale_motion(rel, scale)
        f, _ = oracle.build_motion_maps(hr)
        s = rng.integers(-1, 2, (rows, cols, ch)).astype(np.float32)
        signs.append(s)
        fwd_floats.append(f)
        fm, _ = ops.build_maps(dev(rel), scale)
        fwd_maps.append(fm)
        signs_dev.append(dev(ops.pack_signs(s)))
    with np.errstate(all="ignore"):
        ref = _backproject_ref(X, signs, fwd_floats, scale, gk, B, w, tau, lam)
    got = host(ops.backproject_update(dev(X), fwd_maps, signs_dev, scale, gk, B, w, tau, lam))
    assert np.isnan(ref).sum() >= 1 and np.isinf(ref).sum() >= 2
    assert np.array_equal(ref, got, equal_nan=True)


@pytest.mark.parametrize("ch,border", [(1, 7), (3, 7), (3, 0), (1, 3)])
def test_crop_convert(ops, rng, ch, border):
    X = (rng.random((40, 52, ch)) * 300 - 20).astype(np.float32)
    X[20, 20] = 0.5  # ties round to even
    X[20, 21] = 1.5
    X[20, 22] = 254.5
    X[20, 23] = np.nan
    inner = X[border:X.shape[0] - border, border:X.shape[1] - border]
    assert np.array_equal(host(ops.crop_convert(dev(X), border, False)), inner, equal_nan=True)
    assert np.array_equal(host(ops.crop_convert(dev(X), border, True)), oracle.convert_8u(inner))
