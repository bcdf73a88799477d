"""Host-side logic of libsrx.so that needs no GPU: the C ABI loads and exports what include/srx.h declares,
parameter table, ring/window/span helpers and the streaming driver's bookkeeping (dry-run handle)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from oracle import stream as ostream
from superresolution_amd import _lib as L, api, spans

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "srx.h")).read()
    declared = set(re.findall(r"SRX_API\s+[\w\s\*]+?\b(srx_\w+)\s*\(", hdr))
    assert len(declared) >= 35
    lib = L.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/srx.h but not exported"
    assert declared == set(L.SIGNATURES), "the ctypes table and include/srx.h disagree"


def test_defaults_match_reference_constructor():
    """lib/src/btv_l1.cpp:277-295 and :461-464"""
    sr = api.BTV_L1(-1)
    assert [sr.getInt(n) for n in ("scale", "iterations", "btvKernelSize", "blurKernelSize", "temporalAreaRadius")] \
        == [4, 180, 7, 5, 4]
    assert [sr.getDouble(n) for n in ("tau", "lambda", "alpha", "blurSigma")] == [1.3, 0.03, 0.7, 0.0]


def test_set_get_and_validation():
    sr = api.BTV_L1(-1)
    sr.set("scale", 2)
    sr.set("iterations", 10)
    sr.set("tau", 0.5)
    sr.set("lambda", 0)  # int on a double parameter converts, lambda = 0 switches the regulariser off
    sr.set("blurSigma", 1.5)
    assert sr.getInt("scale") == 2 and sr.getInt("iterations") == 10
    assert sr.getDouble("tau") == 0.5 and sr.getDouble("lambda") == 0.0 and sr.getDouble("blurSigma") == 1.5
    # what the reference only debug-asserts (lib/src/btv_l1.cpp:359-365) is rejected, and the old value kept
    for name, bad in [("scale", 1), ("iterations", 0), ("tau", 0.0), ("alpha", -1.0), ("btvKernelSize", 0),
                      ("blurKernelSize", 4), ("blurKernelSize", 0), ("blurSigma", -0.1), ("temporalAreaRadius", -1),
                      ("btvKernelSize", 17), ("temporalAreaRadius", 20)]:
        with pytest.raises(L.SrxError) as e:
            sr.set(name, bad)
        assert e.value.code == L.SRX_ERR_INVALID
    assert sr.getInt("scale") == 2 and sr.getDouble("tau") == 0.5
    with pytest.raises(L.SrxError):
        sr.set("noSuchParameter", 1)
    with pytest.raises(L.SrxError):
        sr.getInt("tau")  # wrong type


def test_flow_parameters():
    sr = api.BTV_L1(-1)
    lib = L.lib()
    d, i = C.c_double(), C.c_int()
    for name, val in [("tau", 0.25), ("lambda", 0.15), ("theta", 0.3), ("epsilon", 0.01)]:
        L.check(lib.srx_flow_get_double(sr.handle, name.encode(), C.byref(d)))
        assert d.value == val
    for name, val in [("nscales", 5), ("warps", 5), ("iterations", 300)]:
        L.check(lib.srx_flow_get_int(sr.handle, name.encode(), C.byref(i)))
        assert i.value == val
    sr.flowSet("warps", 3)
    L.check(lib.srx_flow_get_int(sr.handle, b"warps", C.byref(i)))
    assert i.value == 3
    with pytest.raises(L.SrxError):
        sr.flowSet("warps", 0)
    # iterations per kernel launch: 0 (chosen by the size of the level) unless a kernel is asked for by name
    L.check(lib.srx_flow_get_int(sr.handle, b"iterationsPerLaunch", C.byref(i)))
    assert i.value == 0
    for ok in (1, 4, 8, 0):
        sr.flowSet("iterationsPerLaunch", ok)
        L.check(lib.srx_flow_get_int(sr.handle, b"iterationsPerLaunch", C.byref(i)))
        assert i.value == ok
    for bad in (2, 3, 16, -1):
        with pytest.raises(L.SrxError):
            sr.flowSet("iterationsPerLaunch", bad)


def test_ring_slot_and_window_match_oracle():
    lib = L.lib()
    for size in (1, 3, 9, 17):
        for idx in range(-40, 60):
            assert lib.srx_ring_slot(idx, size) == oracle.border_wrap(idx, size)
    s, e, b = C.c_int(), C.c_int(), C.c_int()
    for r in (0, 1, 4, 8):
        for store in range(2 * r, 2 * r + 30):
            for idx in range(0, store + 1):
                lib.srx_window(idx, r, store, C.byref(s), C.byref(e), C.byref(b))
                assert (s.value, e.value, b.value) == ostream.window(idx, r, store)


def test_host_kernels_tables():
    from superresolution_amd import ops

    for B, a in [(7, 0.7), (5, 0.5), (1, 0.7), (15, 0.9)]:
        assert np.array_equal(ops.calc_btv_weights(B, a), oracle.calc_btv_weights(B, a))
    for n, s in [(1, 0), (3, 0), (5, 0), (7, 0), (9, 0), (5, 1.2), (31, 6.0)]:
        assert np.array_equal(ops.gaussian_kernel(n, s), oracle.gaussian_kernel(n, s))
    with pytest.raises(L.SrxError):
        ops.gaussian_kernel(4, 0)


def test_span_plan_covers_stream_once():
    for total, world, r in [(100, 8, 4), (17, 4, 2), (16, 8, 1), (30, 1, 4), (33, 2, 8), (40, 8, 4)]:
        seen = []
        for rank in range(world):
            a, b, i0, i1 = spans.span_plan(total, r, world, rank)
            seen += list(range(a, b + 1))
            assert i0 == max(a - r, 0) and i1 == min(b + r, total - 1)
            assert (b - a + 1) in (total // world, total // world + 1)
        assert seen == list(range(total))
    with pytest.raises(L.SrxError):
        spans.span_plan(10, 4, 2, 2)
    # The ranks at the ends of the stream have one halo only: with `radius` frames of their own their handles
    # could not warm up (2 * radius + 1 reads), so the plan is refused for every rank alike; radius + 1 is enough.
    for rank in range(8):
        with pytest.raises(L.SrxError):
            spans.span_plan(32, 4, 8, rank)
        spans.span_plan(40, 4, 8, rank)
    with pytest.raises(L.SrxError):
        spans.span_plan(8, 1, 8, 3)


def test_edge_spans_of_radius_plus_one_frames_stream_and_shorter_ones_cannot():
    """Dry-run handles fed as a job of 8 ranks would feed them: spans of radius + 1 own frames warm up on every
    rank; a rank at the end of the stream with only `radius` own frames runs out of frames in its warm-up."""
    r = 4

    def run(own, left, right):
        sr = api.BTV_L1(-1)
        sr.set("temporalAreaRadius", r)
        sr.setOpticalFlow(lambda a, b: np.zeros((6, 8, 2), np.float32))
        f = [np.zeros((6, 8, 3), np.uint8)] * (left + own + right)
        spans.configure_span(sr, f[:left], f[left:left + own], f[left + own:])
        n = 0
        while sr.nextFrame() is not None:
            n += 1
        return n

    assert run(r + 1, 0, r) == r + 1 and run(r + 1, r, r) == r + 1 and run(r + 1, r, 0) == r + 1
    with pytest.raises(L.SrxError):
        run(r, 0, r)


def _events(sr):
    names = {L.SRX_EV_READ: "read", L.SRX_EV_FLOW: "flow", L.SRX_EV_SOLVE: "solve", L.SRX_EV_OUT: "out"}
    out = []
    for t, a, b, c, d in sr.drainEvents():
        out.append({"read": ("read",), "flow": ("flow", a), "solve": ("solve", a, b, c, d), "out": ("out", a)}[
            names[t]])
    return out


@pytest.mark.parametrize("n,r", [(12, 4), (9, 4), (7, 2), (5, 2), (3, 1), (4, 0), (30, 4), (18, 8)])
def test_dry_run_bookkeeping_matches_oracle_driver(n, r, monkeypatch):
    """Same reads, flows, solves (window, base) and outputs, in the same order, as BTV_L1::initImpl/processImpl."""
    frames = [np.full((6, 8, 3), i, np.uint8) for i in range(n)]
    zero = np.zeros((6, 8, 2), np.float32)
    monkeypatch.setattr(ostream, "btvl1_process", lambda src, rel, base, p: np.zeros((10, 18, 3), np.float32))
    o = ostream.BTV_L1(lambda a, b: zero)
    o.temporalAreaRadius = r
    o.setFrameSource(ostream.ListFrameSource(frames))
    n_ref = 0
    while o.nextFrame() is not None:
        n_ref += 1

    sr = api.BTV_L1(-1)
    sr.set("temporalAreaRadius", r)
    sr.set("traceEvents", 1)
    sr.setOpticalFlow(lambda a, b: zero)
    sr.setFrameSource(ostream.ListFrameSource(frames))
    n_got = 0
    while True:
        f = sr.nextFrame()
        if f is None:
            break
        assert f == (6 * 4 - 14, 8 * 4 - 14, 3)  # K10: (s*rows - 2B) x (s*cols - 2B)
        n_got += 1
    assert n_got == n_ref == n
    assert _events(sr) == o.log


def test_stream_too_short_and_reset():
    frames = [np.zeros((6, 8), np.uint8) for _ in range(5)]
    sr = api.BTV_L1(-1)
    sr.setOpticalFlow(lambda a, b: np.zeros((6, 8, 2), np.float32))
    src = ostream.ListFrameSource(frames)
    sr.setFrameSource(src)
    with pytest.raises(L.SrxError) as e:  # r = 4 needs 9 frames (CV_Assert at lib/src/btv_l1.cpp:479)
        sr.nextFrame()
    assert e.value.code == L.SRX_ERR_SOURCE
    sr.set("temporalAreaRadius", 2)
    sr.reset()  # rewinds the source and re-arms the warm-up (lib/src/super_resolution.cpp:137-141)
    assert src.pos == 0
    count = 0
    while sr.nextFrame() is not None:
        count += 1
    assert count == 5
    assert sr.nextFrame() is None
    # no source at all
    sr2 = api.BTV_L1(-1)
    with pytest.raises(L.SrxError):
        sr2.nextFrame()


def test_frame_type_change_is_rejected():
    frames = [np.zeros((6, 8), np.uint8)] * 3 + [np.zeros((6, 9), np.uint8)] * 3
    sr = api.BTV_L1(-1)
    sr.set("temporalAreaRadius", 1)
    sr.setOpticalFlow(lambda a, b: np.zeros((6, 8, 2), np.float32))
    sr.setFrameSource(ostream.ListFrameSource(frames))
    with pytest.raises(L.SrxError):
        while sr.nextFrame() is not None:
            pass


def test_default_flow_is_farneback_and_a_removed_flow_is_an_error():
    """opticalFlow = new Farneback (lib/src/btv_l1.cpp:287): a new object streams without a flow being set."""
    sr = api.BTV_L1(-1)
    sr.set("temporalAreaRadius", 1)
    sr.set("traceEvents", 1)
    sr.setFrameSource(ostream.ListFrameSource([np.zeros((6, 8), np.uint8)] * 4))
    n = 0
    while sr.nextFrame() is not None:
        n += 1
    assert n == 4
    L.check(sr._lib.srx_set_flow_builtin(sr._h, L.SRX_FLOW_NONE))
    sr.reset()
    sr.setFrameSource(ostream.ListFrameSource([np.zeros((6, 8), np.uint8)] * 4))
    with pytest.raises(L.SrxError) as e:
        sr.nextFrame()
    assert e.value.code == L.SRX_ERR_FLOW


def test_output_range_restricts_solves_to_the_span():
    """A span [a, b] fed with its halo solves exactly the owned frames, with the windows of the global stream."""
    n, r = 24, 3
    frames = [np.full((6, 8), i, np.uint8) for i in range(n)]
    zero = np.zeros((6, 8, 2), np.float32)

    def run(first_frame, last_frame, own_first, own_last):
        sr = api.BTV_L1(-1)
        sr.set("temporalAreaRadius", r)
        sr.set("traceEvents", 1)
        sr.setOpticalFlow(lambda a, b: zero)
        sr.setFrameSource(ostream.ListFrameSource(frames[first_frame:last_frame + 1]))
        sr.setOutputRange(own_first - first_frame, own_last - first_frame)
        outs = 0
        while sr.nextFrame() is not None:
            outs += 1
        ev = _events(sr)
        solves = [(e[1] + first_frame, e[2] + first_frame, e[3] + first_frame, e[4]) for e in ev if e[0] == "solve"]
        return outs, solves

    whole_outs, whole = run(0, n - 1, 0, n - 1)
    assert whole_outs == n and len(whole) == n
    got = []
    world = 3
    for rank in range(world):
        a, b, i0, i1 = spans.span_plan(n, r, world, rank)
        outs, solves = run(i0, i1, a, b)
        assert outs == b - a + 1
        got += solves
    assert sorted(got) == sorted(whole)


def test_dry_handle_walks_process_and_rejects_compute_calls():
    sr = api.BTV_L1(-1)
    frames = [np.zeros((6, 8), np.float32)] * 3
    # process on a dry-run handle: arguments are checked, nothing is computed
    sr.process(frames, [np.zeros((6, 8, 2), np.float32)] * 2, 0)
    with pytest.raises(L.SrxError):
        sr.process(frames, [np.zeros((6, 9, 2), np.float32)] * 2, 0)
    # without motions the flow plug-in is asked once per frame other than the base (lib/src/btv_l1.cpp:311-321)
    calls = []
    sr.setOpticalFlow(lambda a, b: calls.append(1) or np.zeros((6, 8, 2), np.float32))
    sr.process(frames, None, 1)
    assert len(calls) == 2
    with pytest.raises(L.SrxError):
        sr.calcFlow(frames[0], frames[1])
    # pairs side by side need a GPU handle too; the number in flight is a checked parameter of the handle
    import ctypes as C
    with pytest.raises(L.SrxError, match="GPU handle"):
        L.check(L.lib().srx_flow_calc_device_pairs(sr.handle, 0, None, None, 6, 8, None, None, 2))
    assert sr.getInt("flowPairsInFlight") == 3
    sr.set("flowPairsInFlight", 1)
    with pytest.raises(L.SrxError, match="flowPairsInFlight"):
        sr.set("flowPairsInFlight", 0)
    assert sr.getInt("flowPairsInFlight") == 1
    assert L.lib().srx_get_kernel_path() == 0
