"""Python mirror of the reference's SuperResolution / BTV_L1 interface over the C ABI.

Names follow lib/include/super_resolution.hpp:146-195 (BTV_L1, setFrameSource,
reset, nextFrame, set/getInt/getDouble and the parameter names registered at
lib/src/super_resolution.cpp:83-93) plus the spellings the same code took in
OpenCV's superres module (createSuperResolution_BTVL1, setInput).  All
arithmetic happens in libsrx.so on the GPU; this file only marshals buffers.

Frames are numpy arrays (host) or torch CUDA tensors (device), uint8 or
float32, shaped (rows, cols) or (rows, cols, channels).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Sequence

import numpy as np

from . import _lib as L

_INT_PARAMS = ("scale", "iterations", "btvKernelSize", "blurKernelSize", "temporalAreaRadius", "outputMemory",
               "traceEvents", "kernelTiming", "readAhead", "flowPairsInFlight")
_DOUBLE_PARAMS = ("tau", "lambda", "alpha", "blurSigma")


def _is_torch(a) -> bool:
    return type(a).__module__.startswith("torch")


def _describe(a, keep: list) -> L.SrxImage:
    """Image header for a numpy array or a torch tensor; `keep` pins temporaries."""
    im = L.SrxImage()
    if _is_torch(a):
        import torch

        if not a.is_contiguous():
            a = a.contiguous()
        keep.append(a)
        im.data = a.data_ptr()
        im.mem = L.SRX_MEM_DEVICE if a.is_cuda else L.SRX_MEM_HOST
        if a.is_cuda:
            # the library works on a stream of its own: whatever torch still has queued for this tensor must
            # have run before the pointer crosses the C ABI (which takes finished buffers, like a cv::Mat)
            torch.cuda.current_stream(a.device).synchronize()
        if a.dtype == torch.uint8:
            im.depth = L.SRX_8U
        elif a.dtype == torch.float32:
            im.depth = L.SRX_32F
        elif a.dtype in (torch.uint16, torch.int16):  # int16 storage is read as the 16-bit unsigned depth
            im.depth = L.SRX_16U
        else:
            raise TypeError("frames must be uint8, uint16 or float32")
        shape = tuple(a.shape)
        esz = a.element_size()
    else:
        a = np.asarray(a)
        if a.dtype not in (np.uint8, np.uint16, np.float32):
            raise TypeError("frames must be uint8, uint16 or float32")
        if not a.flags["C_CONTIGUOUS"]:
            a = np.ascontiguousarray(a)
        keep.append(a)
        im.data = a.ctypes.data
        im.mem = L.SRX_MEM_HOST
        im.depth = {np.dtype(np.uint8): L.SRX_8U, np.dtype(np.uint16): L.SRX_16U}.get(a.dtype, L.SRX_32F)
        shape = a.shape
        esz = a.itemsize
    if len(shape) == 2:
        shape = (shape[0], shape[1], 1)
    im.rows, im.cols, im.channels = shape
    im.step = shape[1] * shape[2] * esz
    return im


class DenseOpticalFlow:
    """lib/include/optical_flow.hpp:41-46.  Subclass and implement calc(frame0, frame1) -> (rows, cols, 2)
    float32 (numpy or torch CUDA tensor)."""

    def calc(self, frame0, frame1):
        raise NotImplementedError

    def collectGarbage(self):
        pass


class PlanarOpticalFlow(DenseOpticalFlow):
    """The two-output form calc(frame0, frame1, flow1, flow2) of the same interface (lib/src/optical_flow.cpp:
    71-81): subclass and implement calc2(frame0, frame1) -> (u, v), two (rows, cols) float32 planes."""

    def calc2(self, frame0, frame1):
        raise NotImplementedError


class CallableFlow(DenseOpticalFlow):
    def __init__(self, fn: Callable):
        self.fn = fn

    def calc(self, frame0, frame1):
        return self.fn(frame0, frame1)


class DualTVL1(DenseOpticalFlow):
    """Marker for the built-in GPU TV-L1 flow; parameters are set through BTV_L1.flowSet()."""


class Farneback(DenseOpticalFlow):
    """The built-in GPU Farneback flow with the parameters of the reference's class Farneback
    (lib/src/optical_flow.cpp:41-50): pyrScale, numLevels, winSize, numIters, polyN, polySigma, flags.
    Attributes set on this object are applied when it is handed to BTV_L1.setOpticalFlow()."""

    _NAMES = ("pyrScale", "numLevels", "winSize", "numIters", "polyN", "polySigma", "flags")

    def __init__(self, **kw):
        self.pyrScale, self.numLevels, self.winSize, self.numIters = 0.5, 5, 13, 10
        self.polyN, self.polySigma, self.flags = 5, 1.1, 0
        for k, v in kw.items():
            if k not in self._NAMES:
                raise TypeError("unknown Farneback parameter: " + k)
            setattr(self, k, v)


def _view(im: L.SrxImage, device: int = 0):
    """numpy (host) or torch (device `device`) view of a library-owned image."""
    shape = (im.rows, im.cols, im.channels)
    n = im.rows * im.cols * im.channels
    if im.mem == L.SRX_MEM_HOST:
        dt = {L.SRX_8U: np.uint8, L.SRX_16U: np.uint16}.get(im.depth, np.float32)
        buf = (C.c_ubyte * (n * np.dtype(dt).itemsize)).from_address(im.data)
        a = np.frombuffer(buf, dtype=dt).reshape(shape)
        return a[:, :, 0] if im.channels == 1 else a
    import torch

    # zero-copy view of device memory through the CUDA array interface
    class _Dev:
        pass

    d = _Dev()
    d.__cuda_array_interface__ = {
        "shape": shape,
        "typestr": {L.SRX_8U: "|u1", L.SRX_16U: "<u2"}.get(im.depth, "<f4"),
        "data": (im.data, False),
        "version": 2,
    }
    t = torch.as_tensor(d, device=torch.device("cuda", max(device, 0)))
    return t[:, :, 0] if im.channels == 1 else t


class BTV_L1:
    """SuperResolution.BTV_L1 (lib/include/super_resolution.hpp:165-195)."""

    def __init__(self, device: int = 0):
        self._lib = L.lib()
        h = C.c_void_p()
        L.check(self._lib.srx_create_btvl1(device, C.byref(h)))
        self._h = h
        self._device = device
        self._keep = []
        self._src = None
        self._flow = Farneback()  # opticalFlow = new Farneback, lib/src/btv_l1.cpp:287 (the library's default)
        self._cb = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.srx_destroy(h)

    # -- Algorithm::set / get (lib/src/super_resolution.cpp:83-93)
    def set(self, name: str, value):
        if name == "opticalFlow":
            return self.setOpticalFlow(value)
        if name in _DOUBLE_PARAMS:
            L.check(self._lib.srx_set_double(self._h, name.encode(), float(value)))
        else:
            L.check(self._lib.srx_set_int(self._h, name.encode(), int(value)))

    def getInt(self, name: str) -> int:
        v = C.c_int()
        L.check(self._lib.srx_get_int(self._h, name.encode(), C.byref(v)))
        return v.value

    def getDouble(self, name: str) -> float:
        v = C.c_double()
        L.check(self._lib.srx_get_double(self._h, name.encode(), C.byref(v)))
        return v.value

    def flowSet(self, name: str, value):
        if isinstance(value, float):
            L.check(self._lib.srx_flow_set_double(self._h, name.encode(), value))
        else:
            L.check(self._lib.srx_flow_set_int(self._h, name.encode(), int(value)))

    def setOpticalFlow(self, flow):
        """set("opticalFlow", ...) of samples/quality_test/main.cpp:267."""
        if isinstance(flow, DualTVL1):
            self._flow = flow
            L.check(self._lib.srx_set_flow_builtin(self._h, L.SRX_FLOW_TVL1))
            return
        if isinstance(flow, Farneback):
            self._flow = flow
            L.check(self._lib.srx_set_flow_builtin(self._h, L.SRX_FLOW_FARNEBACK))
            for n in Farneback._NAMES:
                v = getattr(flow, n)
                self.flowSet(n, float(v) if n in ("pyrScale", "polySigma") else int(v))
            return
        if callable(flow) and not isinstance(flow, DenseOpticalFlow):
            flow = CallableFlow(flow)
        self._flow = flow
        state = {"keep": []}
        dev = self._device
        if isinstance(flow, PlanarOpticalFlow):

            def _cb2(_user, f0, f1, out_u, out_v):
                try:
                    u, v = flow.calc2(_view(f0.contents, dev), _view(f1.contents, dev))
                    keep = []
                    hu, hv = _describe(u, keep), _describe(v, keep)
                    if hu.channels != 1 or hv.channels != 1 or hu.depth != L.SRX_32F or hv.depth != L.SRX_32F:
                        return 1
                    state["keep"] = keep  # valid until the next call
                    C.memmove(out_u, C.byref(hu), C.sizeof(L.SrxImage))
                    C.memmove(out_v, C.byref(hv), C.sizeof(L.SrxImage))
                    return 0
                except Exception:  # a Python exception must not unwind through C
                    import traceback

                    traceback.print_exc()
                    return 1

            self._flow_cb = L.FLOW2_FN(_cb2)
            L.check(self._lib.srx_set_flow_callback2(self._h, self._flow_cb, None))
            return

        def _cb(_user, f0, f1, out):
            try:
                a0, a1 = _view(f0.contents, dev), _view(f1.contents, dev)
                res = flow.calc(a0, a1)
                keep = []
                hdr = _describe(res, keep)
                if hdr.channels != 2 or hdr.depth != L.SRX_32F:
                    return 1
                state["keep"] = keep  # valid until the next call
                C.memmove(out, C.byref(hdr), C.sizeof(L.SrxImage))
                return 0
            except Exception:  # a Python exception must not unwind through C
                import traceback

                traceback.print_exc()
                return 1

        self._flow_cb = L.FLOW_FN(_cb)
        L.check(self._lib.srx_set_flow_callback(self._h, self._flow_cb, None))

    # -- SuperResolution (lib/src/super_resolution.cpp:131-152)
    def setFrameSource(self, source):
        """source: object with nextFrame() -> array or None, and reset()."""
        self._src = source
        state = {"keep": []}

        def _next(_user, out):
            try:
                f = source.nextFrame()
                if f is None:
                    return 0
                keep = []
                hdr = _describe(f, keep)
                state["keep"] = keep
                C.memmove(out, C.byref(hdr), C.sizeof(L.SrxImage))
                return 1
            except Exception:
                import traceback

                traceback.print_exc()
                return 0

        def _reset(_user):
            source.reset()

        self._next_cb = L.NEXT_FRAME_FN(_next)
        self._reset_cb = L.RESET_FN(_reset)
        L.check(self._lib.srx_set_input(self._h, self._next_cb, self._reset_cb, None))

    setInput = setFrameSource  # upstream superres spelling

    def reset(self):
        L.check(self._lib.srx_reset(self._h))

    def setOutputRange(self, first: int, last: int):
        L.check(self._lib.srx_set_output_range(self._h, first, last))

    def nextFrame(self):
        """Next super-resolved frame (uint8), or None at the end of the stream.  The array aliases a buffer
        owned by the object and is valid until the next call (as the reference's returned Mat)."""
        im = L.SrxImage()
        L.check(self._lib.srx_next_frame(self._h, C.byref(im)))
        if im.rows == 0:
            return None
        if not im.data:  # dry-run handle: sizes only
            return (im.rows, im.cols, im.channels)
        return _view(im, self._device)

    def drainEvents(self, max_events: int = 4096):
        buf = (C.c_int * (5 * max_events))()
        n = self._lib.srx_drain_events(self._h, buf, max_events)
        return [tuple(buf[5 * i:5 * i + 5]) for i in range(n)]

    # -- BTV_L1_Base::process (lib/src/btv_l1.cpp:297-355)
    def process(self, src: Sequence, forwardMotions: Optional[Sequence] = None, baseIdx: int = 0, out=None):
        keep = []
        n = len(src)
        frames = (L.SrxImage * n)(*[_describe(f, keep) for f in src])
        motions = None
        if forwardMotions is not None:
            if len(forwardMotions) != n - 1:
                raise ValueError("forwardMotions must have len(src) - 1 entries")
            motions = (L.SrxImage * max(n - 1, 1))(*[_describe(m, keep) for m in forwardMotions])
        s, B = self.getInt("scale"), self.getInt("btvKernelSize")
        f0 = frames[0]
        shape = (f0.rows * s - 2 * B, f0.cols * s - 2 * B, f0.channels)
        if self._device < 0:
            # dry-run handle: arguments are checked and the flow plug-in is called, nothing is computed; the
            # result is the output's shape, as nextFrame() gives sizes only on such a handle
            L.check(self._lib.srx_process(self._h, frames, n, motions, baseIdx,
                                          C.byref(_describe(np.empty(shape, np.uint8 if f0.depth == L.SRX_8U else
                                                                     np.uint16 if f0.depth == L.SRX_16U else np.float32),
                                                            keep))))
            return shape
        if out is None:
            if f0.mem == L.SRX_MEM_DEVICE:
                import torch

                out = torch.empty(shape, device=torch.device("cuda", max(self._device, 0)),
                                  dtype={L.SRX_8U: torch.uint8, L.SRX_16U: torch.uint16}.get(f0.depth, torch.float32))
            else:
                out = np.empty(shape, {L.SRX_8U: np.uint8, L.SRX_16U: np.uint16}.get(f0.depth, np.float32))
        oh = _describe(out, keep)
        L.check(self._lib.srx_process(self._h, frames, n, motions, baseIdx, C.byref(oh)))
        if f0.channels == 1 and np.ndim(src[0]) == 2:
            out = out[:, :, 0]
        return out

    def calcFlow(self, frame0, frame1, planar: bool = False):
        """DenseOpticalFlow::calc through the object's flow plug-in (lib/include/optical_flow.hpp:44)."""
        keep = []
        f0, f1 = _describe(frame0, keep), _describe(frame1, keep)
        if planar:
            u = np.empty((f0.rows, f0.cols), np.float32)
            v = np.empty((f0.rows, f0.cols), np.float32)
            hu, hv = _describe(u, keep), _describe(v, keep)
            L.check(self._lib.srx_flow_calc(self._h, C.byref(f0), C.byref(f1), C.byref(hu), C.byref(hv)))
            return u, v
        fl = np.empty((f0.rows, f0.cols, 2), np.float32)
        hf = _describe(fl, keep)
        L.check(self._lib.srx_flow_calc(self._h, C.byref(f0), C.byref(f1), C.byref(hf), None))
        return fl

    def calcFlowDevice(self, gray0, gray1):
        """Built-in flow (TV-L1, or Farneback when that is set) on two CUDA float grey images; returns ((rows, cols, 2) CUDA tensor, executed iterations)."""
        import torch

        rows, cols = gray0.shape
        flow = torch.empty((rows, cols, 2), dtype=torch.float32, device=gray0.device)
        n = C.c_longlong()
        torch.cuda.synchronize(gray0.device)
        L.check(self._lib.srx_flow_calc_device(self._h, gray0.contiguous().data_ptr(), gray1.contiguous().data_ptr(),
                                               rows, cols, flow.data_ptr(), C.byref(n)))
        return flow, n.value

    def calcFlowDevicePairs(self, gray0, gray1, in_flight: int = 2):
        """Built-in flow on n independent pairs of CUDA float grey images (lists of equal-sized tensors), up to
        `in_flight` pairs side by side; returns (list of (rows, cols, 2) CUDA tensors, list of executed iterations)."""
        import torch

        n = len(gray0)
        if len(gray1) != n:
            raise ValueError("calcFlowDevicePairs: gray0 and gray1 differ in length")
        if n == 0:
            return [], []
        rows, cols = gray0[0].shape
        a = [g.contiguous() for g in gray0]
        b = [g.contiguous() for g in gray1]
        if any(t.shape != (rows, cols) or t.dtype != torch.float32 for t in a + b):
            raise ValueError("calcFlowDevicePairs: all images must be float32 of one size")
        flows = [torch.empty((rows, cols, 2), dtype=torch.float32, device=a[0].device) for _ in range(n)]
        P = C.c_void_p * n
        its = (C.c_longlong * n)()
        torch.cuda.synchronize(a[0].device)
        L.check(self._lib.srx_flow_calc_device_pairs(self._h, n, P(*[t.data_ptr() for t in a]),
                                                     P(*[t.data_ptr() for t in b]), rows, cols,
                                                     P(*[t.data_ptr() for t in flows]), its, in_flight))
        return flows, list(its)

    def collectGarbage(self):
        L.check(self._lib.srx_collect_garbage(self._h))

    def synchronize(self):
        L.check(self._lib.srx_synchronize(self._h))

    def kernelTimes(self):
        """(forward_ms, backproject_ms, iterations) summed since kernelTiming was last set."""
        f, b, n = C.c_double(), C.c_double(), C.c_long()
        L.check(self._lib.srx_kernel_times(self._h, C.byref(f), C.byref(b), C.byref(n)))
        return f.value, b.value, n.value

    @property
    def handle(self):
        return self._h


def createSuperResolution_BTVL1(device: int = 0) -> BTV_L1:
    """Upstream cv::superres factory spelling of `new BTV_L1` (samples/video/main.cpp:106)."""
    return BTV_L1(device)
