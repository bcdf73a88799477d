"""Two-process (gloo, CPU) test of the span partitioning: halo exchange between neighbouring ranks and the
per-span bookkeeping.  The arithmetic needs a GPU, so the handles here are dry-run handles; the GPU variant
(tests/test_gpu_spans.py) checks the pixels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_PER_RANK, RADIUS = 7, 2


def _frame(t):
    g = np.random.default_rng(1000 + t)
    return g.integers(0, 256, (6, 8, 3), dtype=np.uint8)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from superresolution_amd import _lib as L, api, spans

        total = N_PER_RANK * world
        a, b, i0, i1 = spans.span_plan(total, RADIUS, world, rank)
        own = [torch.from_numpy(_frame(t)) for t in range(a, b + 1)]
        left, right = spans.exchange_halo(own, RADIUS)
        ok = len(left) == a - i0 and len(right) == i1 - b
        for i, f in enumerate(left):
            ok &= bool(np.array_equal(f.numpy(), _frame(i0 + i)))
        for i, f in enumerate(right):
            ok &= bool(np.array_equal(f.numpy(), _frame(b + 1 + i)))

        sr = api.BTV_L1(-1)
        sr.set("temporalAreaRadius", RADIUS)
        sr.set("traceEvents", 1)
        sr.setOpticalFlow(lambda x, y: np.zeros((6, 8, 2), np.float32))
        spans.configure_span(sr, [f.numpy() for f in left], [f.numpy() for f in own], [f.numpy() for f in right])
        outs = 0
        while sr.nextFrame() is not None:
            outs += 1
        solves = [(e[1] + i0, e[2] + i0, e[3] + i0, e[4]) for e in sr.drainEvents() if e[0] == L.SRX_EV_SOLVE]
        q.put((rank, ok, outs, solves))
    finally:
        dist.destroy_process_group()


def test_two_rank_halo_exchange_and_span_windows():
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = N_PER_RANK * world
    got = []
    for rank, ok, outs, solves in res:
        assert ok, f"rank {rank}: halo frames differ from the neighbour's"
        assert outs == N_PER_RANK
        got += solves
    # union of the per-rank solves = the unpartitioned stream's solves (global window function)
    from oracle import stream as ostream

    want = []
    for idx in range(total):
        store = min(max(idx + RADIUS, 2 * RADIUS), total - 1)
        start, end, base = ostream.window(idx, RADIUS, store)
        want.append((idx, start, end, base))
    assert sorted(got) == want


def _pair_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from superresolution_amd import spans

        total_pairs = 11  # frames 0..11
        first, last = spans.pair_shard(total_pairs, world, rank)
        # every rank reads frames [first, last] and "computes" pair t from frames (t, t + 1): a checksum stands in
        # for the flow so that the gathered result can be compared with the unsharded loop
        frames = {t: _frame(t) for t in range(first, last + 1)}
        mine = torch.tensor([[t, int(frames[t].sum()) - int(frames[t + 1].sum())] for t in range(first, last)],
                            dtype=torch.int64).reshape(-1, 2)
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([mine.shape[0]]))
        # result hand-over only (outside the data path): ragged gather by padding to the longest shard
        longest = max(int(n) for n in sizes)
        padded = torch.zeros((longest, 2), dtype=torch.int64)
        padded[:mine.shape[0]] = mine
        out = [torch.zeros_like(padded) for _ in range(world)]
        dist.all_gather(out, padded)
        parts = [o[:int(n)] for o, n in zip(out, sizes)]
        q.put((rank, first, last, torch.cat(parts).tolist()))
    finally:
        dist.destroy_process_group()


def test_two_rank_pair_sharding_of_the_flow_stage():
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pair_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [[t, int(_frame(t).sum()) - int(_frame(t + 1).sum())] for t in range(11)]
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 6, 6, 11)
    for _rank, _a, _b, gathered in res:
        assert gathered == want


def test_pair_shards_cover_every_pair_once():
    from superresolution_amd import spans

    for total in (0, 1, 5, 8, 17):
        for world in (1, 2, 3, 8):
            cuts = [spans.pair_shard(total, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1
    with pytest.raises(ValueError):
        spans.pair_shard(4, 2, 2)


def _refusal_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from superresolution_amd import spans

        # the last rank holds `radius` frames only: too few for an edge span.  The middle rank's own span is fine,
        # yet it must refuse as well instead of waiting for a neighbour that never posts its messages.
        n_own = RADIUS if rank == world - 1 else RADIUS + 2
        own = [torch.from_numpy(_frame(10 * rank + t)) for t in range(n_own)]
        try:
            spans.exchange_halo(own, RADIUS)
            refused = ""
        except ValueError as e:
            refused = str(e)
        # and with spans every rank accepts, the exchange posted early and waited for later gives the same halos
        own = [torch.from_numpy(_frame(10 * rank + t)) for t in range(RADIUS + 1)]
        ex = spans.HaloExchange(own, RADIUS)
        busy = sum(int(f.sum()) for f in own)       # something to do while the frames travel
        left, right = ex.wait()
        ok = busy > 0 and len(left) == (RADIUS if rank > 0 else 0) and len(right) == (RADIUS if rank < world - 1 else 0)
        for i, f in enumerate(left):
            ok &= bool(np.array_equal(f.numpy(), _frame(10 * (rank - 1) + 1 + i)))
        for i, f in enumerate(right):
            ok &= bool(np.array_equal(f.numpy(), _frame(10 * (rank + 1) + i)))
        q.put((rank, refused, ok))
    finally:
        dist.destroy_process_group()


def test_a_span_that_is_too_short_is_refused_on_every_rank_alike():
    world = 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_refusal_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, refused, ok in res:
        assert "rank 2 holds 2 frames" in refused, (rank, refused)
        assert ok, f"rank {rank}: halos of the in-flight exchange differ"
