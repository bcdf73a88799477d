"""Span-partitioned streams reproduce the unpartitioned stream bit for bit (one process, one GPU, one handle per
simulated rank)."""
import numpy as np
import pytest

from oracle import stream as ostream

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_equals_whole(world):
    from superresolution_amd import api, spans, synth

    n, r, rows, cols, ch, T = 13, 2, 12, 16, 3, 5
    lr, flows, _ = synth.make_sequence(n, rows, cols, ch, 4, seed=77)
    lr = [f.numpy() for f in lr]
    fl = [f.numpy() for f in flows]

    def run(first, left, own, right):
        it = iter(fl[first:])
        sr = api.BTV_L1()
        sr.set("iterations", T)
        sr.set("temporalAreaRadius", r)
        sr.setOpticalFlow(lambda a, b: next(it))
        spans.configure_span(sr, left, own, right)
        outs = []
        while True:
            f = sr.nextFrame()
            if f is None:
                break
            outs.append(np.array(f, copy=True))
        return outs

    whole = run(0, [], lr, [])
    assert len(whole) == n
    got = []
    for rank in range(world):
        a, b, i0, i1 = spans.span_plan(n, r, world, rank)
        got += run(i0, lr[i0:a], lr[a:b + 1], lr[b + 1:i1 + 1])
    assert len(got) == n
    for x, y in zip(whole, got):
        assert np.array_equal(x, y)
