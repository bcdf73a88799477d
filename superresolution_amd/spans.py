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


def exchange_halo(own: Sequence[torch.Tensor], radius: int, rank: Optional[int] = None, world: Optional[int] = None,
                  group=None) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """Swap span-boundary frames with the neighbouring ranks.

    `own` holds this rank's frames in stream order (all ranks: same shape and dtype; at least `radius` frames,
    and `radius` + 1 on the first and the last rank, which have one halo only and whose streams otherwise could
    not warm up - `span_plan` refuses such partitions).
    Returns (left_halo, right_halo): the last `radius` frames of rank-1 and the first `radius` frames of rank+1
    (empty lists at the ends of the stream).  Point-to-point only; no collective touches the data path.
    """
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    if radius == 0 or world == 1:
        return [], []
    edge = rank == 0 or rank == world - 1
    if len(own) < radius + (1 if edge else 0):
        raise ValueError("every span must hold at least `radius` frames, the first and the last `radius` + 1")
    left = [torch.empty_like(own[0]) for _ in range(radius)] if rank > 0 else []
    right = [torch.empty_like(own[0]) for _ in range(radius)] if rank < world - 1 else []
    ops = []
    for i in range(radius):
        if rank > 0:
            ops.append(dist.P2POp(dist.isend, own[i].contiguous(), rank - 1, group))
            ops.append(dist.P2POp(dist.irecv, left[i], rank - 1, group))
        if rank < world - 1:
            ops.append(dist.P2POp(dist.isend, own[len(own) - radius + i].contiguous(), rank + 1, group))
            ops.append(dist.P2POp(dist.irecv, right[i], rank + 1, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    return left, right


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
