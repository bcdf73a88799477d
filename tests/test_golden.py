"""Golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py from the oracle).

They pin the oracle (CPU test) and the GPU path (gpu test) against regressions.  They are this repository's
own vectors: the reference ships none and cannot be built here, so they do not prove reference parity.
"""
import os

import numpy as np
import pytest

import oracle
from oracle import stream as ostream

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SOLVES = ["solve_rgb_s4", "solve_gray_s2", "solve_rgb_sigma"]


def load_solve(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    v = z["params"]
    p = oracle.BtvParams(int(v[0]), int(v[1]), float(v[2]), float(v[3]), float(v[4]), int(v[5]), int(v[6]), float(v[7]))
    return z, p


@pytest.mark.parametrize("name", SOLVES)
def test_oracle_reproduces_golden_solve(name):
    z, p = load_solve(name)
    y = [f.astype(np.float32) for f in z["frames"]]
    out, full = oracle.btvl1_process(y, list(z["flows"]), int(z["base"]), p, return_full=True)
    assert np.array_equal(out, z["out"]) and np.array_equal(full, z["x_full"])


def test_oracle_reproduces_golden_stream():
    z = np.load(os.path.join(GOLD, "stream_rgb_r2.npz"))
    it = iter(list(z["flows"]))
    o = ostream.BTV_L1(lambda a, b: next(it))
    o.params = oracle.BtvParams(iterations=int(z["iterations"]))
    o.temporalAreaRadius = int(z["radius"])
    o.setFrameSource(ostream.ListFrameSource(list(z["frames"])))
    for ref in z["outs"]:
        assert np.array_equal(o.nextFrame(), ref)
    assert o.nextFrame() is None


@pytest.mark.gpu
@pytest.mark.parametrize("path", [0, 1], ids=["auto", "generic"])
@pytest.mark.parametrize("name", SOLVES)
def test_gpu_reproduces_golden_solve(name, path):
    import torch
    from superresolution_amd import api, ops

    z, p = load_solve(name)
    ops.set_kernel_path(path)
    try:
        sr = api.BTV_L1()
        for k, v in [("scale", p.scale), ("iterations", p.iterations), ("tau", p.tau), ("lambda", p.lambda_),
                     ("alpha", p.alpha), ("btvKernelSize", p.btv_kernel_size), ("blurKernelSize", p.blur_kernel_size),
                     ("blurSigma", p.blur_sigma)]:
            sr.set(k, v)
        y = [torch.from_numpy(f.astype(np.float32)).cuda() for f in z["frames"]]
        fl = [torch.from_numpy(f).cuda() for f in z["flows"]]
        out, full = ops.process_device(sr, y, fl, int(z["base"]), return_full=True)
        assert np.array_equal(out.cpu().numpy(), z["out"]) and np.array_equal(full.cpu().numpy(), z["x_full"])
    finally:
        ops.set_kernel_path(0)


@pytest.mark.gpu
def test_gpu_reproduces_golden_stream():
    from superresolution_amd import api

    z = np.load(os.path.join(GOLD, "stream_rgb_r2.npz"))
    it = iter(list(z["flows"]))
    sr = api.BTV_L1()
    sr.set("iterations", int(z["iterations"]))
    sr.set("temporalAreaRadius", int(z["radius"]))
    sr.setOpticalFlow(lambda a, b: next(it))
    sr.setFrameSource(ostream.ListFrameSource(list(z["frames"])))
    for ref in z["outs"]:
        assert np.array_equal(np.asarray(sr.nextFrame()), ref)
    assert sr.nextFrame() is None


def _flow_case():
    from oracle import tvl1

    z = np.load(os.path.join(GOLD, "flow_pair_rgb.npz"))
    n, w, it = (int(v) for v in z["tvl1_params"])
    return z, tvl1.Tvl1Params(nscales=n, warps=w, iterations=it), dict(nscales=n, warps=w, iterations=it)


def test_oracle_reproduces_golden_flows():
    from oracle import farneback, tvl1

    z, tp, _ = _flow_case()
    a, b = z["frames"]
    assert np.array_equal(tvl1.calc(a, b, tp), z["tvl1"])
    assert np.array_equal(farneback.calc(a, b), z["farneback"])


@pytest.mark.gpu
def test_gpu_reproduces_golden_flows():
    from superresolution_amd import api

    z, _, kw = _flow_case()
    a, b = z["frames"]
    sr = api.BTV_L1()
    assert np.array_equal(sr.calcFlow(a, b), z["farneback"])  # the flow of a new object (lib/src/btv_l1.cpp:287)
    sr.setOpticalFlow(api.DualTVL1())
    for k, v in kw.items():
        sr.flowSet(k, v)
    assert np.array_equal(sr.calcFlow(a, b), z["tvl1"])
    sr.setOpticalFlow(api.Farneback())
    assert np.array_equal(sr.calcFlow(a, b), z["farneback"])
