"""Multi-GPU partitioning of a stream by contiguous spans of output frames.

Output frame idx depends only on the low-resolution frames and forward motions
of its temporal window (BTV_L1::processFrame, lib/src/btv_l1.cpp:527-549), so
the stream shards with no reduction of any kind: rank g owns outputs [a, b] and
needs input frames [a - r, b + r].  The only exchange is the halo of r
low-resolution frames on each side of a span boundary, sent point to point
between neighbouring ranks (RCCL over xGMI on GPUs, gloo on CPU in tests).

The reference itself is single-process; this module is the new multi-GPU layer.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib as L


def span_plan(total_frames: int, radius: int, world: int, rank: int) -> Tuple[int, int, int, int]:
    """(out_first, out_last, in_first, in_last) of `rank`; srx_span_plan of the C ABI."""
    v = [C.c_int() for _ in range(4)]
    L.check(L.lib().srx_span_plan(total_frames, radius, world, rank, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def _check_span_lengths(n_own: int, like: torch.Tensor, radius: int, rank: int, world: int, group) -> None:
    """Every rank learns every span length (one int per rank, a control message outside the data path) and applies
    the same rule to all of them, so that either all ranks go on to the exchange or all of them raise: a rank that
    refused on its own would leave its neighbours waiting for messages that never come."""
    mine = torch.tensor([n_own], dtype=torch.int64, device=like.device)
    lengths = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(lengths, mine, group=group)
    for g, n in enumerate(int(t.item()) for t in lengths):
        edge = g == 0 or g == world - 1
        if n < radius + (1 if edge else 0):
            raise ValueError(f"rank {g} holds {n} frames: every span must hold at least `radius` = {radius} frames, "
                             f"the first and the last `radius` + 1 (span_plan refuses such partitions)")


class HaloExchange:
    """The halo swap of one rank in flight: post() starts the sends and receives, wait() returns the halos.
    Work that does not need the neighbours' frames (synthesis, uploads, the first windows of an inner span's
    warm-up) can run between the two."""

    def __init__(self, own: Sequence[torch.Tensor], radius: int, rank: Optional[int] = None,
                 world: Optional[int] = None, group=None, check: bool = True):
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.left: List[torch.Tensor] = []
        self.right: List[torch.Tensor] = []
        self._reqs = []
        self._keep = []
        rank, world = self.rank, self.world
        if radius == 0 or world == 1:
            return
        if check:
            _check_span_lengths(len(own), own[0], radius, rank, world, group)
        self.left = [torch.empty_like(own[0]) for _ in range(radius)] if rank > 0 else []
        self.right = [torch.empty_like(own[0]) for _ in range(radius)] if rank < world - 1 else []
        ops = []
        for i in range(radius):
            if rank > 0:
                ops.append(dist.P2POp(dist.isend, own[i].contiguous(), rank - 1, group))
                ops.append(dist.P2POp(dist.irecv, self.left[i], rank - 1, group))
            if rank < world - 1:
                ops.append(dist.P2POp(dist.isend, own[len(own) - radius + i].contiguous(), rank + 1, group))
                ops.append(dist.P2POp(dist.irecv, self.right[i], rank + 1, group))
        self._keep = [op.tensor for op in ops]      # the send buffers stay alive until wait()
        self._reqs = dist.batch_isend_irecv(ops)

    def wait(self) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
        for req in self._reqs:
            req.wait()
        self._reqs, self._keep = [], []
        return self.left, self.right


def exchange_halo(own: Sequence[torch.Tensor], radius: int, rank: Optional[int] = None, world: Optional[int] = None,
                  group=None) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """Swap span-boundary frames with the neighbouring ranks.

    `own` holds this rank's frames in stream order (all ranks: same shape and dtype; at least `radius` frames,
    and `radius` + 1 on the first and the last rank, which have one halo only and whose streams otherwise could
    not warm up - `span_plan` refuses such partitions, and so does this function, on every rank alike).
    Returns (left_halo, right_halo): the last `radius` frames of rank-1 and the first `radius` frames of rank+1
    (empty lists at the ends of the stream).  The frames travel point to point; the only collective is the
    gather of one integer per rank (the span lengths) that makes the refusal uniform.
    """
    return HaloExchange(own, radius, rank, world, group).wait()


def pair_shard(total_pairs: int, world: int, rank: int) -> Tuple[int, int]:
    """[first, last) of the frame pairs rank `rank` computes flow for when the flow stage runs on its own.

    Pair t is the motion frame t -> frame t + 1 (opticalFlow->calc(prevFrame, curFrame), lib/src/btv_l1.cpp:522);
    pairs are independent, so the shards are contiguous, differ by at most one pair and need no exchange beyond
    the one frame two neighbouring shards both read (rank g reads frames [first, last])."""
    if world < 1 or not 0 <= rank < world or total_pairs < 0:
        raise ValueError("pair_shard: need world >= 1, 0 <= rank < world, total_pairs >= 0")
    base, extra = divmod(total_pairs, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


class ListSource:
    """IFrameSource over a list of frames (None = end of stream)."""

    def __init__(self, frames):
        self.frames = list(frames)
        self.pos = 0

    def reset(self):
        self.pos = 0

    def nextFrame(self):
        if self.pos >= len(self.frames):
            return None
        f = self.frames[self.pos]
        self.pos += 1
        return f


def configure_span(sr, left_halo: Sequence, own: Sequence, right_halo: Sequence) -> ListSource:
    """Feed `sr` the span plus its halos and restrict its outputs to the owned frames.

    Frame indices inside the handle count from the first frame it is fed, so the owned range is
    [len(left_halo), len(left_halo) + len(own) - 1].  With an empty left halo the stream start's
    warm-up windows apply, with an empty right halo its shrinking tail windows, exactly as in the
    unpartitioned stream.
    """
    src = ListSource(list(left_halo) + list(own) + list(right_halo))
    sr.setFrameSource(src)
    sr.setOutputRange(len(left_halo), len(left_halo) + len(own) - 1)
    return src
