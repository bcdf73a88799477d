"""Built-in TV-L1 flow: oracle sanity (CPU) and GPU-vs-oracle parity.

The reference has no TV-L1 (its flows wrap OpenCV's Farneback etc.), so there is nothing of the reference to
compare with: oracle/tvl1_oracle.c is the specification, the GPU kernels are held to it bit for bit, and the
oracle itself is checked against the known motion of the synthetic sequences."""
import numpy as np
import pytest

from oracle import tvl1
from superresolution_amd import synth


def _pair(rows, cols, ch, seed):
    lr, flows, _ = synth.make_sequence(2, rows, cols, ch, 4, seed=seed)
    return lr[0].numpy(), lr[1].numpy(), flows[0].numpy()


def test_oracle_recovers_known_motion():
    a, b, true = _pair(96, 128, 1, 3)
    est, n = tvl1.calc_gray(tvl1.to_gray(a), tvl1.to_gray(b))
    epe = np.sqrt(((est - true) ** 2).sum(-1))
    assert epe[8:-8, 8:-8].mean() < 0.1
    assert 0 < n <= 5 * 5 * 300


def test_oracle_zero_flow_for_identical_frames():
    a, _, _ = _pair(40, 48, 3, 5)
    est, n = tvl1.calc_gray(tvl1.to_gray(a), tvl1.to_gray(a))
    assert not est.any()
    assert n == 2 * 5  # 2 levels (40x48, 20x24; 10x12 is below 16) x 5 warps x 1 iteration (error 0 at once)


def test_gray_conversion_rules():
    rgb = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8)
    g = tvl1.to_gray(rgb)
    want = [(255 * 1868 + 8192) >> 14, (255 * 9617 + 8192) >> 14, (255 * 4899 + 8192) >> 14,
            (10 * 1868 + 20 * 9617 + 30 * 4899 + 8192) >> 14]
    assert list(g[0]) == want
    f = np.array([[0.0, 0.5, 1.0, 2.0, -1.0]], np.float32)
    assert list(tvl1.to_gray(f)[0]) == [0, 128, 255, 255, 0]  # round half to even: 127.5 -> 128
    assert np.array_equal(tvl1.to_gray(rgb[..., :1]), rgb[..., 0].astype(np.float32))


def test_pyramid_and_gradient_properties(rng):
    img = (rng.random((21, 30)) * 255).astype(np.float32)
    d = tvl1.pyr_down(img)
    assert d.shape == (11, 15)
    assert np.allclose(tvl1.pyr_down(np.full((20, 20), 9.0, np.float32)), 9.0)
    ramp = np.tile(np.arange(12, dtype=np.float32) * 3, (5, 1))
    dx, dy = tvl1.centered_gradient(ramp)
    assert np.array_equal(dx[:, 1:-1], np.full((5, 10), 3.0, np.float32))
    assert np.array_equal(dx[:, 0], np.full(5, 1.5, np.float32)) and not dy.any()
    up = tvl1.resize_linear(ramp, 10, 24)
    assert up.shape == (10, 24) and np.all(np.diff(up[0]) >= 0)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,ch,dtype,kw", [
    (64, 80, 1, np.uint8, {}),
    (48, 72, 3, np.uint8, {}),
    (33, 47, 3, np.uint8, dict(nscales=3, warps=2, iterations=40)),
    (40, 40, 1, np.float32, dict(nscales=2, warps=3, iterations=25, epsilon=0.0005)),
    (20, 36, 4, np.uint8, dict(nscales=5, warps=2, iterations=10)),
])
def test_gpu_flow_matches_oracle_bit_exact(rows, cols, ch, dtype, kw):
    from superresolution_amd import api

    c3 = 3 if ch == 4 else ch
    a, b, _ = _pair(rows, cols, c3, 11 + rows)
    if ch == 4:
        a = np.concatenate([a, a[..., :1]], axis=-1)
        b = np.concatenate([b, b[..., :1]], axis=-1)
    if dtype == np.float32:
        a = (a.astype(np.float32) / 255.0)
        b = (b.astype(np.float32) / 255.0)
    p = tvl1.Tvl1Params(**kw)
    ref = tvl1.calc(a, b, p)
    sr = api.BTV_L1()
    sr.setOpticalFlow(api.DualTVL1())
    for k, v in kw.items():
        sr.flowSet(k, v)
    got = sr.calcFlow(a, b)
    assert np.array_equal(ref, got)
    u, v = sr.calcFlow(a, b, planar=True)
    assert np.array_equal(ref[..., 0], u) and np.array_equal(ref[..., 1], v)


@pytest.mark.gpu
@pytest.mark.parametrize("per_launch", [1, 4, 8, 12, 14])
@pytest.mark.parametrize("rows,cols,kw", [
    (64, 80, {}),                                             # width a multiple of 4: 16-byte single-iteration kernel
    (33, 47, dict(nscales=3, warps=2, iterations=40)),        # odd width: the one-pixel-per-thread kernel
    (150, 264, dict(nscales=3, warps=2, iterations=60)),      # several tiles of every kernel in both directions
    (41, 132, dict(nscales=2, warps=2, iterations=30)),       # a tile column of 4 pixels behind a full one
])
def test_every_iteration_kernel_matches_the_oracle(rows, cols, kw, per_launch):
    """The library picks the iteration kernel by the size and the width of the pyramid level: 8 iterations per
    launch on small levels, 2 or 4 on regions of whole 512-byte rows above 200 000 pixels where the width is a
    multiple of 4 (4 on 32 x 32 tiles or one per launch otherwise).  Images an oracle run can afford only ever reach
    the first, so iterationsPerLaunch forces each kernel onto every level (12 and 14: the whole-row kernel with 2
    and 4 iterations; levels of other widths fall back to 32 x 32 tiles): flows and executed iteration counts must
    be the oracle's with all of them."""
    import torch
    from superresolution_amd import api

    a, b, _ = _pair(rows, cols, 1, 5 + rows)
    ga, gb = tvl1.to_gray(a), tvl1.to_gray(b)
    ref, n_ref = tvl1.calc_gray(ga, gb, tvl1.Tvl1Params(**kw))
    sr = api.BTV_L1()
    sr.setOpticalFlow(api.DualTVL1())
    for k, v in kw.items():
        sr.flowSet(k, v)
    sr.flowSet("iterationsPerLaunch", per_launch)
    flow, n = sr.calcFlowDevice(torch.from_numpy(ga).cuda(), torch.from_numpy(gb).cuda())
    assert n == n_ref
    assert np.array_equal(flow.cpu().numpy(), ref)
    # a second call starts from the first one's predictions of the iteration counts (launch batches): same result
    flow2, n2 = sr.calcFlowDevice(torch.from_numpy(ga).cuda(), torch.from_numpy(gb).cuda())
    assert n2 == n_ref and torch.equal(flow, flow2)


@pytest.mark.gpu
def test_gpu_flow_device_entry_and_iteration_count():
    import torch
    from superresolution_amd import api

    a, b, true = _pair(96, 128, 1, 3)
    ga, gb = tvl1.to_gray(a), tvl1.to_gray(b)
    ref, n_ref = tvl1.calc_gray(ga, gb)
    sr = api.BTV_L1()
    sr.setOpticalFlow(api.DualTVL1())
    flow, n = sr.calcFlowDevice(torch.from_numpy(ga).cuda(), torch.from_numpy(gb).cuda())
    assert n == n_ref
    assert np.array_equal(flow.cpu().numpy(), ref)
    epe = np.sqrt(((ref - true) ** 2).sum(-1))
    assert epe[8:-8, 8:-8].mean() < 0.1


@pytest.mark.gpu
def test_pairs_side_by_side_are_the_flows_of_the_oracle():
    """srx_flow_calc_device_pairs: independent pairs on streams and workspaces of their own, several in flight.
    Every flow and iteration count is the oracle's for its pair, whatever the number in flight; a pair list may be
    empty; a number in flight outside 1 .. 4 is refused."""
    import torch
    from superresolution_amd import api, _lib as L

    lr, _, _ = synth.make_sequence(6, 97, 131, 1, 4, seed=17)
    gray = [tvl1.to_gray(f.numpy()) for f in lr]
    want = [tvl1.calc_gray(gray[t], gray[t + 1]) for t in range(5)]
    dev = [torch.from_numpy(g).cuda() for g in gray]
    sr = api.BTV_L1()
    sr.setOpticalFlow(api.DualTVL1())
    for in_flight in (3, 1, 4, 2):
        flows, its = sr.calcFlowDevicePairs(dev[:-1], dev[1:], in_flight=in_flight)
        assert [int(i) for i in its] == [int(n) for _, n in want]
        for f, (ref, _) in zip(flows, want):
            assert np.array_equal(f.cpu().numpy(), ref)
    # the work of the whole call is what the single calls add up to
    import ctypes as C
    total = C.c_longlong()
    L.check(L.lib().srx_flow_pixel_iterations(sr.handle, C.byref(total)))
    singles = 0
    for t in range(5):
        sr.calcFlowDevice(dev[t], dev[t + 1])
        w = C.c_longlong()
        L.check(L.lib().srx_flow_pixel_iterations(sr.handle, C.byref(w)))
        singles += w.value
    assert total.value == singles > 0
    assert sr.calcFlowDevicePairs([], []) == ([], [])
    for bad in (0, 5):
        with pytest.raises(L.SrxError, match="in_flight"):
            sr.calcFlowDevicePairs(dev[:2], dev[1:3], in_flight=bad)
    # Farneback pairs run one after the other and report no iterations
    fb = api.BTV_L1()
    fb.setOpticalFlow(api.Farneback())
    flows, its = fb.calcFlowDevicePairs(dev[:2], dev[1:3], in_flight=2)
    one, _ = fb.calcFlowDevice(dev[1], dev[2])
    assert its == [0, 0] and torch.equal(flows[1], one)


@pytest.mark.gpu
@pytest.mark.parametrize("mem", ["host", "device"])
def test_batch_process_computes_the_windows_flows_side_by_side(mem):
    """process(frames, None, base) with the built-in TV-L1: the flows frame i -> base (lib/src/btv_l1.cpp:311-321)
    are independent pairs and run flowPairsInFlight at a time; the result is the oracle's for the oracle's flows,
    whatever that number."""
    import torch
    import oracle
    from superresolution_amd import api, _lib as L

    lr, _, _ = synth.make_sequence(5, 40, 56, 3, 4, seed=23)
    frames = [f.numpy() for f in lr]
    base = 2
    rel = [np.zeros((40, 56, 2), np.float32) if i == base else
           tvl1.calc_gray(tvl1.to_gray(frames[i]), tvl1.to_gray(frames[base]))[0] for i in range(5)]
    p = oracle.BtvParams(iterations=4)
    want = oracle.btvl1_run([f.astype(np.float32) for f in frames], rel, base, p)
    want8 = oracle.convert_8u(want)
    sr = api.BTV_L1()
    sr.set("iterations", 4)
    sr.setOpticalFlow(api.DualTVL1())
    src = frames if mem == "host" else [torch.from_numpy(f).cuda() for f in frames]
    assert sr.getInt("flowPairsInFlight") == 3
    for k in (3, 1, 4):
        sr.set("flowPairsInFlight", k)
        got = sr.process(src, None, base)
        got = got.cpu().numpy() if hasattr(got, "cpu") else got
        assert np.array_equal(got, want8), k
    with pytest.raises(L.SrxError, match="flowPairsInFlight"):
        sr.set("flowPairsInFlight", 5)
    assert sr.getInt("flowPairsInFlight") == 4


@pytest.mark.gpu
def test_streaming_with_builtin_flow_matches_oracle():
    """End to end: TV-L1 flow feeding the BTV-L1 solver, against the oracle driver with the oracle flow."""
    import oracle
    from oracle import stream as ostream
    from superresolution_amd import api

    n, r, T = 6, 1, 5
    lr, _, _ = synth.make_sequence(n, 24, 32, 3, 4, seed=21)
    lr = [f.numpy() for f in lr]
    p = tvl1.Tvl1Params(nscales=2, warps=2, iterations=20)
    o = ostream.BTV_L1(lambda f0, f1: tvl1.calc(f0, f1, p))
    o.params = oracle.BtvParams(iterations=T)
    o.temporalAreaRadius = r
    o.setFrameSource(ostream.ListFrameSource(lr))
    sr = api.BTV_L1()
    sr.set("iterations", T)
    sr.set("temporalAreaRadius", r)
    sr.setOpticalFlow(api.DualTVL1())
    for k, v in dict(nscales=2, warps=2, iterations=20).items():
        sr.flowSet(k, v)
    sr.setFrameSource(ostream.ListFrameSource(lr))
    for _ in range(n):
        a, b = o.nextFrame(), sr.nextFrame()
        assert np.array_equal(a, np.asarray(b))
    assert o.nextFrame() is None and sr.nextFrame() is None


@pytest.mark.gpu
@pytest.mark.parametrize("mem", ["host", "device"])
def test_read_ahead_gives_the_same_frames(mem):
    """readAhead computes the next pair's flow beside the current solve: same outputs, frame for frame, the same
    end of stream, and a reset() starts over."""
    import torch
    from oracle import stream as ostream
    from superresolution_amd import api, _lib as L

    n, r, T = 9, 2, 4
    lr, _, _ = synth.make_sequence(n, 24, 32, 3, 4, seed=33)
    frames = [f.numpy() for f in lr] if mem == "host" else [f.cuda() for f in lr]

    def run(read_ahead, twice=False):
        sr = api.BTV_L1()
        sr.set("iterations", T)
        sr.set("temporalAreaRadius", r)
        sr.set("readAhead", read_ahead)
        assert sr.getInt("readAhead") == read_ahead
        sr.setOpticalFlow(api.DualTVL1())
        for k, v in dict(nscales=2, warps=2, iterations=20).items():
            sr.flowSet(k, v)
        if mem == "device":
            sr.set("outputMemory", L.SRX_MEM_DEVICE)
        sr.setFrameSource(ostream.ListFrameSource(frames))
        outs = []
        for _ in range(2 if twice else 1):
            outs.append([])
            while True:
                f = sr.nextFrame()
                if f is None:
                    break
                outs[-1].append(f.cpu().numpy().copy() if mem == "device" else np.array(f, copy=True))
            assert sr.nextFrame() is None
            sr.reset()
        return outs

    ref = run(0)[0]
    assert len(ref) == n
    for got in run(1, twice=True):
        assert len(got) == n
        for a, b in zip(ref, got):
            assert np.array_equal(a, b)


def test_gray_conversion_of_16_bit_frames_follows_convert_to():
    """convertToType(frame, CV_8U, 1) on a 16-bit frame: value * (255/65535) in float, rounded half to even
    (lib/src/input_array_utility.cpp:90-103)."""
    v = np.array([[0, 128, 129, 257, 32767, 32896, 65535, 385]], np.uint16)
    g = tvl1.to_gray(v)
    exp = np.rint(v.astype(np.float32) * np.float32(255.0 / 65535.0))
    assert np.array_equal(g, exp)
    assert g[0, 0] == 0 and g[0, -2] == 255 and g[0, 3] == 1


@pytest.mark.gpu
def test_16_bit_frames_stream_and_batch():
    """16-bit frames: unscaled conversion to float for the solver, rescaled to 8 bit for the flow; the stream
    delivers 8-bit frames (btv_l1.cpp:504), process() delivers the depth it was given (btv_l1.cpp:456)."""
    import oracle
    from oracle import stream as ostream
    from superresolution_amd import api

    n, r, T = 5, 1, 4
    lr, flows, _ = synth.make_sequence(n, 20, 28, 3, 4, seed=5)
    lr16 = [(f.numpy().astype(np.uint16) * 3 + 7) for f in lr]     # values up to 772: beyond 8 bits
    p = tvl1.Tvl1Params(nscales=2, warps=2, iterations=15)
    o = ostream.BTV_L1(lambda f0, f1: tvl1.calc(f0, f1, p))
    o.params = oracle.BtvParams(iterations=T)
    o.temporalAreaRadius = r
    o.setFrameSource(ostream.ListFrameSource(lr16))
    sr = api.BTV_L1()
    sr.set("iterations", T)
    sr.set("temporalAreaRadius", r)
    sr.setOpticalFlow(api.DualTVL1())
    for k, v in dict(nscales=2, warps=2, iterations=15).items():
        sr.flowSet(k, v)
    sr.setFrameSource(ostream.ListFrameSource(lr16))
    for _ in range(n):
        a, b = o.nextFrame(), sr.nextFrame()
        assert a.dtype == np.uint8 and np.array_equal(a, np.asarray(b))
    # batch entry with given motions: result in the frames' depth
    fl = [f.numpy() for f in flows]
    ref = oracle.btvl1_process([f.astype(np.float32) for f in lr16], fl, 2, oracle.BtvParams(iterations=T))
    got = sr.process(lr16, forwardMotions=fl, baseIdx=2)
    assert got.dtype == np.uint16
    assert np.array_equal(got, np.clip(np.rint(ref), 0, 65535).astype(np.uint16))
