"""ctypes wrappers of oracle/tvl1_oracle.c (TEST INFRASTRUCTURE ONLY; parity unpinned, see that file)."""
from __future__ import annotations

import ctypes as C

import numpy as np

_fp = C.POINTER(C.c_float)


class Tvl1Params(C.Structure):
    _fields_ = [("tau", C.c_double), ("lambda_", C.c_double), ("theta", C.c_double), ("nscales", C.c_int),
                ("warps", C.c_int), ("epsilon", C.c_double), ("iterations", C.c_int)]

    def __init__(self, tau=0.25, lambda_=0.15, theta=0.3, nscales=5, warps=5, epsilon=0.01, iterations=300):
        super().__init__(tau, lambda_, theta, nscales, warps, epsilon, iterations)


def declare(L):
    L.orc_tvl1_to_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _fp]
    L.orc_pyr_down.argtypes = [_fp, C.c_int, C.c_int, _fp]
    L.orc_centered_gradient.argtypes = [_fp, C.c_int, C.c_int, _fp, _fp]
    L.orc_tvl1_warp.argtypes = [_fp] * 6 + [C.c_int, C.c_int] + [_fp] * 5
    L.orc_tvl1_iteration.argtypes = [_fp] * 10 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
    L.orc_tvl1_iteration.restype = C.c_uint64
    L.orc_resize_linear.argtypes = [_fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int]
    L.orc_tvl1_calc.argtypes = [_fp, _fp, C.c_int, C.c_int, C.POINTER(Tvl1Params), _fp]
    L.orc_tvl1_calc.restype = C.c_longlong


def _lib():
    from . import lib

    return lib()


def _p(a):
    return a.ctypes.data_as(_fp)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def to_gray(frame) -> np.ndarray:
    f = np.ascontiguousarray(frame)
    if f.ndim == 2:
        f = f[:, :, None]
    assert f.dtype in (np.uint8, np.uint16, np.float32)
    kind = {np.dtype(np.uint8): 0, np.dtype(np.float32): 1, np.dtype(np.uint16): 2}[f.dtype]
    out = np.empty(f.shape[:2], np.float32)
    _lib().orc_tvl1_to_gray(f.ctypes.data_as(C.c_void_p), f.shape[0], f.shape[1], f.shape[2], kind, _p(out))
    return out


def pyr_down(src) -> np.ndarray:
    s = _f32(src)
    dst = np.empty(((s.shape[0] + 1) // 2, (s.shape[1] + 1) // 2), np.float32)
    _lib().orc_pyr_down(_p(s), s.shape[0], s.shape[1], _p(dst))
    return dst


def centered_gradient(src):
    s = _f32(src)
    dx, dy = np.empty_like(s), np.empty_like(s)
    _lib().orc_centered_gradient(_p(s), s.shape[0], s.shape[1], _p(dx), _p(dy))
    return dx, dy


def warp(I0, I1, I1x, I1y, u1, u2):
    a = [_f32(x) for x in (I0, I1, I1x, I1y, u1, u2)]
    outs = [np.empty_like(a[0]) for _ in range(5)]
    _lib().orc_tvl1_warp(*[_p(x) for x in a], a[0].shape[0], a[0].shape[1], *[_p(x) for x in outs])
    return outs  # I1w, I1wx, I1wy, grad, rho_c


def iteration(I1wx, I1wy, grad, rho_c, u1, u2, p11, p12, p21, p22, l_t, theta, taut):
    """In place on u1, u2, p11..p22 (float32 contiguous arrays); returns the fixed-point error."""
    rows, cols = u1.shape
    ro = [_f32(x) for x in (I1wx, I1wy, grad, rho_c)]
    return _lib().orc_tvl1_iteration(*[_p(x) for x in ro], _p(u1), _p(u2), _p(p11), _p(p12), _p(p21), _p(p22), rows,
                                     cols, l_t, theta, taut)


def resize_linear(src, drows, dcols) -> np.ndarray:
    s = _f32(src)
    dst = np.empty((drows, dcols), np.float32)
    _lib().orc_resize_linear(_p(s), s.shape[0], s.shape[1], _p(dst), drows, dcols)
    return dst


def calc_gray(I0, I1, params: Tvl1Params | None = None):
    """Flow I0 -> I1 of two grey float images: ((rows, cols, 2) float32, executed inner iterations)."""
    params = params or Tvl1Params()
    a, b = _f32(I0), _f32(I1)
    flow = np.empty(a.shape + (2,), np.float32)
    n = _lib().orc_tvl1_calc(_p(a), _p(b), a.shape[0], a.shape[1], C.byref(params), _p(flow))
    return flow, int(n)


def calc(frame0, frame1, params: Tvl1Params | None = None):
    """DenseOpticalFlow::calc on frames as a source delivers them (8U/32F, 1/3/4 channels)."""
    return calc_gray(to_gray(frame0), to_gray(frame1), params)[0]
