"""Whole-solve and streaming parity of the GPU path against the CPU oracle.

The iteration contains sign() steps and no step decay, so a last-bit difference
in any intermediate can flip a sign and move a 5x5 patch by up to tau; the GPU
kernels therefore follow the oracle's operation order exactly and these tests
demand bit-identical float images, not a tolerance.  (Oracle-vs-reference
parity itself is unpinned: the reference cannot be built without OpenCV.)
"""
import numpy as np
import pytest

import oracle
from oracle import stream as ostream
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


def dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_case(rng, K, rows, cols, ch, scale, seed):
    from superresolution_amd import synth

    lr, flows, _ = synth.make_sequence(K, rows, cols, ch, scale, seed=seed)
    y = [f.numpy().astype(np.float32) for f in lr]
    fl = [f.numpy() for f in flows]
    return y, fl


def set_params(sr, p):
    sr.set("scale", p.scale)
    sr.set("iterations", p.iterations)
    sr.set("tau", p.tau)
    sr.set("lambda", p.lambda_)
    sr.set("alpha", p.alpha)
    sr.set("btvKernelSize", p.btv_kernel_size)
    sr.set("blurKernelSize", p.blur_kernel_size)
    sr.set("blurSigma", p.blur_sigma)


CASES = [
    # K, rows, cols, ch, base, params
    (5, 24, 32, 1, 2, dict(iterations=12)),
    (5, 20, 28, 3, 2, dict(iterations=12)),
    (9, 18, 22, 3, 4, dict(iterations=8)),
    (3, 26, 30, 3, 0, dict(iterations=10, scale=2)),
    (3, 16, 20, 1, 2, dict(iterations=10, scale=3, blur_kernel_size=3)),
    (4, 16, 20, 3, 1, dict(iterations=6, blur_kernel_size=7, blur_sigma=1.1, btv_kernel_size=5, alpha=0.5)),
    (2, 16, 20, 1, 1, dict(iterations=6, lambda_=0.0)),
    (1, 16, 20, 3, 0, dict(iterations=6)),
    (17, 14, 18, 3, 8, dict(iterations=4, scale=2)),
    # scale 2: an HR width that is not a multiple of 4 (one LR column per lane), more than one strip of either
    # forward kernel, one channel
    (3, 26, 31, 3, 0, dict(iterations=6, scale=2)),
    (3, 10, 130, 3, 1, dict(iterations=4, scale=2)),
    (3, 12, 66, 1, 1, dict(iterations=4, scale=2)),
    (3, 12, 65, 1, 1, dict(iterations=4, scale=2)),
    # scale 2 over several tiles of the update kernel in both directions (sign patches staged per tile and frame)
    (5, 48, 80, 3, 2, dict(iterations=6, scale=2)),
    (4, 40, 72, 1, 3, dict(iterations=5, scale=2, lambda_=0.0)),
    # more window frames than one block of the scale-4 forward kernel takes (two passes over the tile),
    # and a strip boundary (more than 63 LR columns)
    (11, 12, 70, 3, 5, dict(iterations=3)),
    (13, 10, 16, 1, 0, dict(iterations=3)),
    # scale 4 with more than 14 frames: the update kernel's packed sign sums are unpacked in mid-loop (its 10-bit
    # fields hold 14 frames' worth); 33 frames are four passes of the forward kernel, 40 the largest window
    # the tiled forward kernel (other scales and blur sizes) over several tiles of 32 x 8 LR pixels per axis
    (3, 20, 70, 3, 1, dict(iterations=3, scale=3)),
    (3, 18, 40, 1, 0, dict(iterations=3, blur_kernel_size=7, blur_sigma=1.1)),
    (2, 20, 36, 3, 1, dict(iterations=3, scale=2, blur_kernel_size=3)),
    (17, 10, 16, 3, 8, dict(iterations=3)),
    (33, 8, 12, 1, 30, dict(iterations=2)),
    (40, 8, 12, 3, 0, dict(iterations=2)),
]


@pytest.mark.parametrize("K,rows,cols,ch,base,kw", CASES)
def test_process_matches_oracle_bit_exact(rng, K, rows, cols, ch, base, kw):
    from superresolution_amd import api, ops

    p = oracle.BtvParams(**kw)
    y, fl = make_case(rng, K, rows, cols, ch, p.scale, seed=K * 100 + rows)
    ref_out, ref_full = oracle.btvl1_process(y, fl, base, p, return_full=True)
    sr = api.BTV_L1()
    set_params(sr, p)
    out, full = ops.process_device(sr, [dev(f) for f in y], [dev(f) for f in fl], base, return_full=True)
    assert np.array_equal(ref_full, full.cpu().numpy())
    assert np.array_equal(ref_out, out.cpu().numpy())


def test_one_object_through_changing_sizes_windows_and_scales(rng):
    """The solver's workspace only grows, and what it holds per frame (maps in several forms, sign planes, tile
    headers) is laid out by the size of the call: a call must not see what an earlier call of another size, window
    or scale left behind.  One object, calls of changing shape, each against the oracle."""
    from superresolution_amd import api, ops

    sr = api.BTV_L1()
    calls = [(3, 12, 16, 3, dict(iterations=3, scale=2)), (5, 40, 72, 3, dict(iterations=3, scale=2)),
             (9, 14, 70, 3, dict(iterations=2)), (2, 10, 12, 1, dict(iterations=3, scale=2)),
             (4, 20, 24, 1, dict(iterations=2, scale=3, blur_kernel_size=3)), (5, 40, 72, 3, dict(iterations=2, scale=2)),
             (3, 8, 8, 3, dict(iterations=2))]
    for n, (K, rows, cols, ch, kw) in enumerate(calls):
        p = oracle.BtvParams(**kw)
        y, fl = make_case(rng, K, rows, cols, ch, p.scale, seed=900 + n)
        ref_out, ref_full = oracle.btvl1_process(y, fl, K // 2, p, return_full=True)
        set_params(sr, p)
        out, full = ops.process_device(sr, [dev(f) for f in y], [dev(f) for f in fl], K // 2, return_full=True)
        assert np.array_equal(ref_full, full.cpu().numpy()), f"call {n}"
        assert np.array_equal(ref_out, out.cpu().numpy()), f"call {n}"


@pytest.mark.parametrize("scale,rows,cols,ch,amp", [(4, 20, 24, 3, 4.0), (2, 20, 24, 3, 4.0), (2, 14, 132, 3, 2.0),
                                                    (2, 14, 132, 1, 9.0), (2, 16, 33, 3, 4.0)])
def test_run_with_arbitrary_relative_motions(rng, scale, rows, cols, ch, amp):
    """run() with motions that leave the image, so the constant-0 remap border is exercised in both directions;
    at scale 2 the motions also break the runs of consecutive source pixels that the forward kernel loads whole
    (one step within a lane's four pixels, several steps, sources on both sides of the image edge)."""
    from superresolution_amd import api, ops

    K = 3
    p = oracle.BtvParams(iterations=10 if cols < 100 else 4, scale=scale)
    y, _ = make_case(rng, K, rows, cols, ch, p.scale, seed=5)
    rel = [smooth_motion(rng, rows, cols, amp=amp) for _ in range(K)]
    rel[1][:] = 0
    rel[0][:4] += 9.0
    ref = oracle.btvl1_run(y, rel, 1, p)
    sr = api.BTV_L1()
    set_params(sr, p)
    out = ops.run(sr, [dev(f) for f in y], [dev(f) for f in rel], 1)
    assert np.array_equal(ref, out.cpu().numpy())


def test_fixed_point_kat(rng):
    """lambda = 0, one frame, zero motion, y = D H X0 evaluated by the same ops: every sign is 0, X never moves."""
    from superresolution_amd import api, ops

    rows, cols, s = 12, 16, 4
    y0 = (rng.random((rows, cols)) * 255).astype(np.float32)
    X0 = oracle.resize_cubic(y0, rows * s, cols * s)
    gk = oracle.gaussian_kernel(5, 0)
    y = oracle.resize_nearest(oracle.sep_filter(X0, gk), rows, cols)
    # X0 is rebuilt from y by the solver, so feed the fixed point through srx_k_* directly
    zero = np.zeros((rows, cols, 2), np.float32)
    fm, bm = ops.build_maps(dev(zero), s)
    sg = ops.forward_sign(dev(X0), bm, dev(y), s, gk)
    assert np.array_equal(sg.cpu().numpy(), ops.pack_signs(np.zeros((rows, cols, 1))))
    w = oracle.calc_btv_weights(7, 0.7)
    X1 = ops.backproject_update(dev(X0), [fm], [sg], s, gk, 7, w, 1.3, 0.0)
    assert np.array_equal(X1.cpu().numpy(), X0)


class TrueFlow:
    """Optical-flow plug-in that returns the known motion of the synthetic sequence, in call order."""

    def __init__(self, flows):
        self.flows = flows
        self.i = 0

    def __call__(self, prev, cur):
        f = self.flows[self.i]
        self.i += 1
        return f


@pytest.mark.parametrize("n_frames,r,ch,mem,shape", [
    (8, 2, 1, "host", None), (7, 1, 3, "host", None), (12, 4, 3, "host", None), (8, 2, 3, "device", None),
    (5, 2, 1, "host", None), (4, 0, 1, "host", None),
    # scale 2 (8-bit frames reach its kernels as bytes), several tiles of the update kernel in both directions;
    # scale 4 over two strips of the forward kernel
    (7, 2, 3, "host", (40, 72, 2)), (6, 1, 1, "device", (24, 36, 2)), (6, 2, 3, "device", (12, 70, 4))])
def test_streaming_matches_oracle(n_frames, r, ch, mem, shape):
    import torch
    from superresolution_amd import api, synth, _lib as L

    rows, cols, s = shape or (16, 20, 4)
    T = 6 if shape is None else 4
    lr, flows, _ = synth.make_sequence(n_frames, rows, cols, ch, s, seed=40 + n_frames)
    lr_np = [f.numpy() if ch == 3 else f.numpy()[:, :, 0] for f in lr]
    fl_np = [f.numpy() for f in flows]

    o = ostream.BTV_L1(TrueFlow(fl_np))
    o.params = oracle.BtvParams(iterations=T, scale=s)
    o.temporalAreaRadius = r
    o.setFrameSource(ostream.ListFrameSource(lr_np))
    ref = []
    while True:
        f = o.nextFrame()
        if f is None:
            break
        ref.append(f.copy())
    assert len(ref) == n_frames

    sr = api.BTV_L1()
    sr.set("iterations", T)
    sr.set("scale", s)
    sr.set("temporalAreaRadius", r)
    sr.set("traceEvents", 1)
    if mem == "device":
        frames = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in lr_np]
        sr.set("outputMemory", L.SRX_MEM_DEVICE)
        sr.setOpticalFlow(TrueFlow([torch.from_numpy(f).cuda() for f in fl_np]))
    else:
        frames = lr_np
        sr.setOpticalFlow(TrueFlow(fl_np))
    sr.setFrameSource(ostream.ListFrameSource(frames))
    got = []
    while True:
        f = sr.nextFrame()
        if f is None:
            break
        got.append(f.cpu().numpy().copy() if mem == "device" else np.array(f, copy=True))
    assert len(got) == len(ref)
    for a, b in zip(ref, got):
        assert a.shape == b.shape and a.dtype == b.dtype == np.uint8
        assert np.array_equal(a, b)
    # same bookkeeping: reads, flows, solves (window, base) and outputs in the same order
    ev = sr.drainEvents()
    names = {L.SRX_EV_READ: "read", L.SRX_EV_FLOW: "flow", L.SRX_EV_SOLVE: "solve", L.SRX_EV_OUT: "out"}
    mine = []
    for t, a, b, c, d in ev:
        mine.append({"read": ("read",), "flow": ("flow", a), "solve": ("solve", a, b, c, d), "out": ("out", a)}[
            names[t]])
    assert mine == o.log


@pytest.mark.parametrize("kind", ["device", "pinned"])
def test_source_and_flow_plugin_that_reuse_one_buffer(kind):
    """A source that refills ONE device (or pinned host) buffer per call, and a flow plug-in that does the same
    with its result: the library reads such memory on its stream, so it has to finish reading before it calls
    the source / the plug-in again (the image 'must stay valid until the next call of next()').  Larger frames
    than elsewhere, so that a copy still in flight would be caught."""
    import torch
    from superresolution_amd import api, synth, _lib as L

    n_frames, rows, cols, ch, r, T = 9, 96, 128, 3, 2, 3
    lr, flows, _ = synth.make_sequence(n_frames, rows, cols, ch, 4, seed=91)
    lr_np = [f.numpy() for f in lr]
    fl_np = [f.numpy() for f in flows]
    o = ostream.BTV_L1(TrueFlow(fl_np))
    o.params = oracle.BtvParams(iterations=T)
    o.temporalAreaRadius = r
    o.setFrameSource(ostream.ListFrameSource(lr_np))
    ref = []
    while (f := o.nextFrame()) is not None:
        ref.append(f.copy())

    if kind == "device":
        fbuf = torch.empty((rows, cols, ch), dtype=torch.uint8, device="cuda")
        mbuf = torch.empty((rows, cols, 2), dtype=torch.float32, device="cuda")
    else:
        fbuf = torch.empty((rows, cols, ch), dtype=torch.uint8).pin_memory()
        mbuf = torch.empty((rows, cols, 2), dtype=torch.float32).pin_memory()

    class OneBufferSource:
        def __init__(self):
            self.i = 0

        def reset(self):
            self.i = 0

        def nextFrame(self):
            if self.i >= n_frames:
                return None
            fbuf.copy_(torch.from_numpy(lr_np[self.i]))  # overwrites the frame handed out by the previous call
            if kind == "device":
                torch.cuda.synchronize()
            self.i += 1
            return fbuf

    class OneBufferFlow:
        def __init__(self):
            self.i = 0

        def __call__(self, a, b):
            mbuf.copy_(torch.from_numpy(fl_np[self.i]))
            if kind == "device":
                torch.cuda.synchronize()
            self.i += 1
            return mbuf

    sr = api.BTV_L1()
    sr.set("iterations", T)
    sr.set("temporalAreaRadius", r)
    sr.setOpticalFlow(OneBufferFlow())
    sr.setFrameSource(OneBufferSource())
    got = []
    while (f := sr.nextFrame()) is not None:
        got.append(np.array(f, copy=True))
    assert len(got) == len(ref)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)


def test_two_output_flow_plugin():
    """calc(frame0, frame1, flow1, flow2): a plug-in that delivers planar u and v (lib/src/optical_flow.cpp:
    71-81) gives the same stream as one that delivers the interleaved field."""
    from superresolution_amd import api, synth

    n_frames, rows, cols, ch, r, T = 6, 16, 20, 3, 1, 5
    lr, flows, _ = synth.make_sequence(n_frames, rows, cols, ch, 4, seed=93)
    lr_np = [f.numpy() for f in lr]
    fl_np = [f.numpy() for f in flows]

    class Planar(api.PlanarOpticalFlow):
        def __init__(self):
            self.i = 0

        def calc2(self, a, b):
            f = fl_np[self.i]
            self.i += 1
            return np.ascontiguousarray(f[:, :, 0]), np.ascontiguousarray(f[:, :, 1])

    def run(flow):
        sr = api.BTV_L1()
        sr.set("iterations", T)
        sr.set("temporalAreaRadius", r)
        sr.setOpticalFlow(flow)
        sr.setFrameSource(ostream.ListFrameSource(lr_np))
        out = []
        while (f := sr.nextFrame()) is not None:
            out.append(np.array(f, copy=True))
        return out

    a, b = run(TrueFlow(fl_np)), run(Planar())
    assert len(a) == len(b) == n_frames
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("s", [4, 2])
def test_batch_process_8u_host(rng, s):
    """srx_process on 8-bit host frames with forward motions: output is 8 bit, cropped by btvKernelSize.  (8-bit
    frames reach the specialised forward kernels of both scales as bytes, float frames as floats: same values.)"""
    from superresolution_amd import api, synth

    K, rows, cols, ch = 5, 16, 20, 3
    lr, flows, _ = synth.make_sequence(K, rows, cols, ch, s, seed=9)
    p = oracle.BtvParams(scale=s, iterations=8)
    y = [f.numpy().astype(np.float32) for f in lr]
    fl = [f.numpy() for f in flows]
    ref = oracle.convert_8u(oracle.btvl1_process(y, fl, 2, p))
    sr = api.BTV_L1()
    set_params(sr, p)
    out = sr.process([f.numpy() for f in lr], fl, 2)
    assert out.dtype == np.uint8 and out.shape == (rows * s - 14, cols * s - 14, ch)
    assert np.array_equal(ref, out)
    # float frames in, float frames out
    outf = sr.process(y, fl, 2)
    assert outf.dtype == np.float32
    assert np.array_equal(oracle.btvl1_process(y, fl, 2, p), outf)


def test_kernel_timing_times_every_eighth_iteration_and_changes_nothing():
    """kernelTiming = 1: every eighth iteration's two launches carry events tied to the kernels' begin and end;
    kernelTimes() adds them up.  The result of the solve is the same with and without."""
    from superresolution_amd import api, synth

    K, rows, cols, ch = 5, 64, 96, 3
    lr, flows, _ = synth.make_sequence(K, rows, cols, ch, 4, seed=31)
    y = [f.numpy().astype(np.float32) for f in lr]
    fl = [f.numpy() for f in flows]
    sr = api.BTV_L1()
    sr.set("iterations", 20)
    plain = sr.process(y, fl, 2)
    sr.set("kernelTiming", 1)
    timed = sr.process(y, fl, 2)
    assert np.array_equal(plain, timed)
    f, b, n = sr.kernelTimes()
    assert n == 3 and 0 < f < 50 and 0 < b < 50  # iterations 0, 8, 16; milliseconds, summed
    sr.process(y, fl, 2)
    f2, b2, n2 = sr.kernelTimes()
    assert n2 == 6 and f2 > f and b2 > b
    sr.set("kernelTiming", 1)  # setting it again starts the sums afresh
    assert sr.kernelTimes() == (0.0, 0.0, 0)


def test_batch_process_with_flow_plugin(rng):
    """process(src, dst, baseIdx) computes flow frame i -> base through the plug-in (lib/src/btv_l1.cpp:297-326)."""
    from superresolution_amd import api

    K, rows, cols = 3, 16, 20
    y = [(rng.random((rows, cols)) * 255).astype(np.float32) for _ in range(K)]
    rel = {i: smooth_motion(rng, rows, cols) for i in range(K)}
    base = 1
    calls = []

    def flow(f0, f1):
        # identify the frame by content
        i = [j for j in range(K) if np.array_equal(np.asarray(f0), y[j])][0]
        assert np.array_equal(np.asarray(f1), y[base])
        calls.append(i)
        return rel[i]

    p = oracle.BtvParams(iterations=5)
    sr = api.BTV_L1()
    set_params(sr, p)
    sr.setOpticalFlow(flow)
    out = sr.process(y, None, base)
    assert calls == [0, 2]
    rels = [rel[0], np.zeros_like(rel[0]), rel[2]]
    assert np.array_equal(oracle.btvl1_run(y, rels, base, p), out)
