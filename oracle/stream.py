"""CPU restatement of the streaming driver BTV_L1 / SuperResolution.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows, statement by
statement, lib/src/btv_l1.cpp:461-549 (BTV_L1::initImpl / processImpl /
addNewFrame / processFrame) and lib/src/super_resolution.cpp:125-152
(SuperResolution::setFrameSource / reset / nextFrame); the arithmetic is done
by the C oracle.  Bookkeeping only, so plain Python is fast enough.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np

from . import BtvParams, border_wrap, btvl1_process, convert_8u


class ListFrameSource:
    """IFrameSource over a list of frames: nextFrame() returns None at the end
    (the empty Mat of videostab::IFrameSource)."""

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


def _at(index: int, items: list):
    """lib/src/ring_buffer.hpp:36-46"""
    return border_wrap(index, len(items))


class BTV_L1:
    """Streaming BTV-L1.  opticalFlow(prev, cur) -> (h, w, 2) float32, the
    DenseOpticalFlow::calc call of lib/src/btv_l1.cpp:522."""

    def __init__(self, opticalFlow: Callable[[np.ndarray, np.ndarray], np.ndarray]):
        self.params = BtvParams()
        self.temporalAreaRadius = 4  # lib/src/btv_l1.cpp:463
        self.opticalFlow = opticalFlow
        self.frameSource = None
        self.firstCall = True
        # event log for host-logic tests: ("read",), ("flow", i), ("solve", idx, start, end, base), ("out", idx)
        self.log: List[tuple] = []

    # -- SuperResolution (lib/src/super_resolution.cpp:131-152)
    def setFrameSource(self, src):
        self.frameSource = src
        self.firstCall = True

    def reset(self):
        self.frameSource.reset()
        self.firstCall = True

    def nextFrame(self) -> Optional[np.ndarray]:
        if self.firstCall:
            self._initImpl()
            self.firstCall = False
        return self._processImpl()

    # -- BTV_L1 (lib/src/btv_l1.cpp:466-549)
    def _initImpl(self):
        r = self.temporalAreaRadius
        cache = 2 * r + 1
        self.frames = [None] * cache
        self.results = [None] * cache
        self.motions = [None] * cache
        self.prevFrame = None
        self.storePos = -1
        for _t in range(-r, r + 1):
            frame = self.frameSource.nextFrame()
            self.log.append(("read",))
            if frame is None:
                raise RuntimeError("CV_Assert(!frame.empty()) failed: lib/src/btv_l1.cpp:479")
            self._addNewFrame(frame)
        for i in range(0, r + 1):
            self._processFrame(i)
        self.procPos = r
        self.outPos = -1

    def _processImpl(self):
        frame = self.frameSource.nextFrame()
        self.log.append(("read",))
        self._addNewFrame(frame)
        if self.procPos < self.storePos:
            self.procPos += 1
            self._processFrame(self.procPos)
        if self.outPos < self.storePos:
            self.outPos += 1
            self.log.append(("out", self.outPos))
            return convert_8u(self.results[_at(self.outPos, self.results)])
        return None

    def _addNewFrame(self, frame):
        if frame is None:
            return
        self.storePos += 1
        self.frames[_at(self.storePos, self.frames)] = np.asarray(frame, dtype=np.float32)
        if self.storePos > 0:
            self.log.append(("flow", self.storePos - 1))
            self.motions[_at(self.storePos - 1, self.motions)] = np.asarray(
                self.opticalFlow(self.prevFrame, frame), dtype=np.float32)
        self.prevFrame = np.array(frame, copy=True)

    def _processFrame(self, idx: int):
        r = self.temporalAreaRadius
        startIdx = max(idx - r, 0)
        procIdx = idx
        endIdx = min(startIdx + 2 * r, self.storePos)
        src, rel = [], []
        baseIdx = -1
        k = 0
        for i in range(startIdx, endIdx + 1):
            if i == procIdx:
                baseIdx = k
            src.append(self.frames[_at(i, self.frames)])
            if i < endIdx:
                rel.append(self.motions[_at(i, self.motions)])
            k += 1
        self.log.append(("solve", idx, startIdx, endIdx, baseIdx))
        self.results[_at(idx, self.results)] = btvl1_process(src, rel, baseIdx, self.params)


def window(idx: int, r: int, store_pos: int):
    """The temporal window of processFrame as a pure function
    (lib/src/btv_l1.cpp:529-531,539-540): returns (start, end, base)."""
    start = max(idx - r, 0)
    end = min(start + 2 * r, store_pos)
    return start, end, idx - start
