"""ctypes wrappers of oracle/farneback_oracle.c (TEST INFRASTRUCTURE ONLY; parity unpinned, see that file)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .tvl1 import to_gray

_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)


class FarnebackParams(C.Structure):
    """Defaults: lib/src/optical_flow.cpp:41-50."""

    _fields_ = [("pyrScale", C.c_double), ("numLevels", C.c_int), ("winSize", C.c_int), ("numIters", C.c_int),
                ("polyN", C.c_int), ("polySigma", C.c_double), ("flags", C.c_int)]

    def __init__(self, pyrScale=0.5, numLevels=5, winSize=13, numIters=10, polyN=5, polySigma=1.1, flags=0):
        super().__init__(pyrScale, numLevels, winSize, numIters, polyN, polySigma, flags)


def declare(L):
    L.orc_farneback_smooth_taps.argtypes = [C.c_double, _fp]
    L.orc_farneback_smooth_taps.restype = C.c_int
    L.orc_farneback_blur.argtypes = [_fp, C.c_int, C.c_int, _fp, C.c_int, _fp]
    L.orc_farneback_poly_setup.argtypes = [C.c_int, C.c_double, _fp, _fp, _fp, _dp]
    L.orc_farneback_poly_setup.restype = C.c_int
    L.orc_farneback_poly_exp.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _dp, _fp]
    L.orc_farneback_update_matrices.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, _fp]
    L.orc_farneback_update_flow.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp]
    L.orc_farneback_update_flow_gaussian.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp]
    L.orc_farneback_levels.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
    L.orc_farneback_levels.restype = C.c_int
    L.orc_farneback_calc.argtypes = [_fp, _fp, C.c_int, C.c_int, C.POINTER(FarnebackParams), _fp]
    L.orc_farneback_calc.restype = C.c_int


def _lib():
    from . import lib

    return lib()


def _p(a):
    return a.ctypes.data_as(_fp)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def smooth_taps(scale: float) -> np.ndarray:
    n = _lib().orc_farneback_smooth_taps(scale, None)
    k = np.empty(n, np.float32)
    _lib().orc_farneback_smooth_taps(scale, _p(k))
    return k


def poly_setup(n: int, sigma: float):
    g, xg, xxg = (np.empty(n + 1, np.float32) for _ in range(3))
    ig = np.empty(4, np.float64)
    rc = _lib().orc_farneback_poly_setup(n, sigma, _p(g), _p(xg), _p(xxg), ig.ctypes.data_as(_dp))
    assert rc == 0
    return g, xg, xxg, ig


def poly_exp(src, n: int = 5, sigma: float = 1.1) -> np.ndarray:
    s = _f32(src)
    g, xg, xxg, ig = poly_setup(n, sigma)
    dst = np.empty(s.shape + (5,), np.float32)
    _lib().orc_farneback_poly_exp(_p(s), s.shape[0], s.shape[1], n, _p(g), _p(xg), _p(xxg), ig.ctypes.data_as(_dp),
                                  _p(dst))
    return dst


def update_matrices(R0, R1, flow) -> np.ndarray:
    a, b, f = _f32(R0), _f32(R1), _f32(flow)
    M = np.empty_like(a)
    _lib().orc_farneback_update_matrices(_p(a), _p(b), _p(f), a.shape[0], a.shape[1], _p(M))
    return M


def update_flow(M, win: int, gaussian: bool = False) -> np.ndarray:
    m = _f32(M)
    flow = np.empty(m.shape[:2] + (2,), np.float32)
    fn = _lib().orc_farneback_update_flow_gaussian if gaussian else _lib().orc_farneback_update_flow
    fn(_p(m), m.shape[0], m.shape[1], win, _p(flow))
    return flow


def calc_gray(I0, I1, params: FarnebackParams | None = None) -> np.ndarray:
    """Flow I0 -> I1 of two grey float images: (rows, cols, 2) float32."""
    params = params or FarnebackParams()
    a, b = _f32(I0), _f32(I1)
    flow = np.empty(a.shape + (2,), np.float32)
    rc = _lib().orc_farneback_calc(_p(a), _p(b), a.shape[0], a.shape[1], C.byref(params), _p(flow))
    if rc != 0:
        raise ValueError("orc_farneback_calc: bad parameters")
    return flow


def calc(frame0, frame1, params: FarnebackParams | None = None) -> np.ndarray:
    """DenseOpticalFlow::calc on frames as a source delivers them (8U/16U/32F, 1/3/4 channels)."""
    return calc_gray(to_gray(frame0), to_gray(frame1), params)
