"""Generates the golden vectors of tests/golden/*.npz from the CPU oracle.

The reference ships no fixtures and cannot be built here (it needs OpenCV), so these vectors are
produced by this repository's own restatement (oracle/).  They pin the GPU path and the oracle
against regressions; they are NOT outputs of the reference binary.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from oracle import stream as ostream  # noqa: E402
from superresolution_amd import synth  # noqa: E402


def solve_case(name, K, rows, cols, ch, base, seed, **kw):
    p = oracle.BtvParams(**kw)
    lr, flows, _ = synth.make_sequence(K, rows, cols, ch, p.scale, seed=seed, rot_growth=True)
    lr = [f.numpy() for f in lr]
    fl = [f.numpy() for f in flows]
    out, full = oracle.btvl1_process([f.astype(np.float32) for f in lr], fl, base, p, return_full=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), frames=np.stack(lr), flows=np.stack(fl), base=base,
                        params=np.array([p.scale, p.iterations, p.tau, p.lambda_, p.alpha, p.btv_kernel_size,
                                         p.blur_kernel_size, p.blur_sigma], np.float64),
                        out=out, x_full=full)


def stream_case(name, n, rows, cols, ch, r, T, seed):
    lr, flows, _ = synth.make_sequence(n, rows, cols, ch, 4, seed=seed, rot_growth=True)
    lr = [f.numpy() for f in lr]
    fl = [f.numpy() for f in flows]
    it = iter(fl)
    o = ostream.BTV_L1(lambda a, b: next(it))
    o.params = oracle.BtvParams(iterations=T)
    o.temporalAreaRadius = r
    o.setFrameSource(ostream.ListFrameSource(lr))
    outs = []
    while True:
        f = o.nextFrame()
        if f is None:
            break
        outs.append(f)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), frames=np.stack(lr), flows=np.stack(fl), radius=r,
                        iterations=T, outs=np.stack(outs))


def flow_case(name, rows, cols, ch, seed):
    """One frame pair and the flows of the two built-in methods (oracle/tvl1_oracle.c, oracle/farneback_oracle.c)."""
    from oracle import farneback, tvl1

    lr, _, _ = synth.make_sequence(2, rows, cols, ch, 4, seed=seed)
    a, b = lr[0].numpy(), lr[1].numpy()
    tp = tvl1.Tvl1Params(nscales=3, warps=3, iterations=30)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), frames=np.stack([a, b]),
                        tvl1_params=np.array([tp.nscales, tp.warps, tp.iterations]), tvl1=tvl1.calc(a, b, tp),
                        farneback=farneback.calc(a, b))


if __name__ == "__main__":
    if sys.argv[1:] == ["flow"]:  # only the flow vectors (the others stay as committed)
        flow_case("flow_pair_rgb", 66, 80, 3, seed=105)
        print("flow vectors written")
        sys.exit(0)
    solve_case("solve_rgb_s4", 5, 16, 20, 3, 2, seed=101, iterations=8)
    solve_case("solve_gray_s2", 3, 18, 22, 1, 0, seed=102, iterations=6, scale=2)
    solve_case("solve_rgb_sigma", 4, 12, 16, 3, 1, seed=103, iterations=5, blur_kernel_size=7, blur_sigma=1.2,
               btv_kernel_size=5, alpha=0.6)
    stream_case("stream_rgb_r2", 7, 12, 16, 3, 2, 5, seed=104)
    flow_case("flow_pair_rgb", 66, 80, 3, seed=105)
    print("golden vectors written")
