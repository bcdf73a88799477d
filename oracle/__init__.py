"""ctypes front end of the CPU oracle (oracle/btvl1_oracle.c, oracle/tvl1_oracle.c, oracle/farneback_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg.  The shipped GPU path never imports this package.

PARITY UNPINNED: see the header of btvl1_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no-op when the library is already up to date)."""
    args = ["make", "-C", _HERE]
    if force:
        args.insert(1, "-B")
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


class BtvParams(C.Structure):
    """Mirror of orc_btv_params; defaults are lib/src/btv_l1.cpp:279-286."""

    _fields_ = [
        ("scale", C.c_int),
        ("iterations", C.c_int),
        ("tau", C.c_double),
        ("lambda_", C.c_double),
        ("alpha", C.c_double),
        ("btv_kernel_size", C.c_int),
        ("blur_kernel_size", C.c_int),
        ("blur_sigma", C.c_double),
    ]

    def __init__(self, scale=4, iterations=180, tau=1.3, lambda_=0.03, alpha=0.7, btv_kernel_size=7,
                 blur_kernel_size=5, blur_sigma=0.0):
        super().__init__(scale, iterations, tau, lambda_, alpha, btv_kernel_size, blur_kernel_size, blur_sigma)


_fp = C.POINTER(C.c_float)
_fpp = C.POINTER(_fp)


def _declare(L):
    L.orc_border_reflect101.argtypes = [C.c_int, C.c_int]
    L.orc_border_reflect101.restype = C.c_int
    L.orc_border_wrap.argtypes = [C.c_int, C.c_int]
    L.orc_border_wrap.restype = C.c_int
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_get_max_threads.restype = C.c_int
    L.orc_calc_relative_motions.argtypes = [_fpp, _fpp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_cubic_table_x.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), _fp]
    L.orc_cubic_table_y.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), _fp]
    L.orc_resize_cubic.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int]
    L.orc_upscale_motion.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp]
    L.orc_build_motion_maps.argtypes = [_fp, C.c_int, C.c_int, _fp, _fp]
    L.orc_map_to_short.argtypes = [_fp, C.c_size_t, C.POINTER(C.c_short)]
    L.orc_remap_nearest.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp]
    L.orc_gaussian_kernel.argtypes = [C.c_int, C.c_double, _fp]
    L.orc_sep_filter.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, _fp]
    L.orc_resize_nearest.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int]
    L.orc_diff_sign.argtypes = [_fp, _fp, C.c_size_t, _fp]
    L.orc_upscale.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]
    L.orc_calc_btv_weights.argtypes = [C.c_int, C.c_double, _fp]
    L.orc_calc_btv_weights.restype = C.c_int
    L.orc_btv_regularization.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]
    L.orc_add.argtypes = [_fp, _fp, C.c_size_t, _fp]
    L.orc_add_weighted.argtypes = [_fp, C.c_double, _fp, C.c_double, C.c_double, C.c_size_t, _fp]
    L.orc_convert_8u.argtypes = [_fp, C.c_size_t, C.POINTER(C.c_ubyte)]
    L.orc_convert_32f.argtypes = [C.POINTER(C.c_ubyte), C.c_size_t, _fp]
    L.orc_btvl1_run.argtypes = [_fpp, _fpp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(BtvParams), _fp,
                                _fp]
    L.orc_btvl1_run.restype = C.c_int
    L.orc_btvl1_process.argtypes = L.orc_btvl1_run.argtypes
    L.orc_btvl1_process.restype = C.c_int
    if hasattr(L, "orc_tvl1_calc"):
        from . import tvl1 as _tvl1

        _tvl1.declare(L)
    if hasattr(L, "orc_farneback_calc"):
        from . import farneback as _fb

        _fb.declare(L)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(_fp)


def _pp(arrs: Sequence[np.ndarray]):
    return (_fp * len(arrs))(*[_p(a) for a in arrs])


def _img3(a) -> np.ndarray:
    a = _f32(a)
    if a.ndim == 2:
        a = a[:, :, None]
    return a


# ---------------------------------------------------------------------------
# thin numpy wrappers, one per oracle function


def border_reflect101(p: int, n: int) -> int:
    return lib().orc_border_reflect101(p, n)


def border_wrap(p: int, n: int) -> int:
    return lib().orc_border_wrap(p, n)


def set_threads(n: int) -> None:
    lib().orc_set_threads(n)


def max_threads() -> int:
    return lib().orc_get_max_threads()


def calc_relative_motions(fwd: Sequence[np.ndarray], base_idx: int, rows: int, cols: int):
    """lib/src/btv_l1.cpp:36-57; fwd has n-1 motions, result has n."""
    fwd = [_f32(f) for f in fwd]
    n = len(fwd) + 1
    rel = [np.empty((rows, cols, 2), np.float32) for _ in range(n)]
    fin = _pp(fwd) if fwd else (_fp * 1)()
    lib().orc_calc_relative_motions(fin, _pp(rel), n, base_idx, rows, cols)
    return rel


def cubic_table_x(scols: int, dcols: int):
    sx = np.empty(dcols, np.int32)
    al = np.empty((dcols, 4), np.float32)
    lib().orc_cubic_table_x(scols, dcols, sx.ctypes.data_as(C.POINTER(C.c_int)), _p(al))
    return sx, al


def cubic_table_y(srows: int, drows: int):
    sy = np.empty(drows, np.int32)
    be = np.empty((drows, 4), np.float32)
    lib().orc_cubic_table_y(srows, drows, sy.ctypes.data_as(C.POINTER(C.c_int)), _p(be))
    return sy, be


def resize_cubic(src, drows: int, dcols: int) -> np.ndarray:
    s = _img3(src)
    dst = np.empty((drows, dcols, s.shape[2]), np.float32)
    lib().orc_resize_cubic(_p(s), s.shape[0], s.shape[1], s.shape[2], _p(dst), drows, dcols)
    return dst if np.ndim(src) == 3 else dst[:, :, 0]


def upscale_motion(lr, scale: int) -> np.ndarray:
    lr = _f32(lr)
    hr = np.empty((lr.shape[0] * scale, lr.shape[1] * scale, 2), np.float32)
    lib().orc_upscale_motion(_p(lr), lr.shape[0], lr.shape[1], scale, _p(hr))
    return hr


def build_motion_maps(motion):
    m = _f32(motion)
    fwd = np.empty_like(m)
    bwd = np.empty_like(m)
    lib().orc_build_motion_maps(_p(m), m.shape[0], m.shape[1], _p(fwd), _p(bwd))
    return fwd, bwd


def map_to_short(fmap) -> np.ndarray:
    m = _f32(fmap)
    out = np.empty(m.shape, np.int16)
    lib().orc_map_to_short(_p(m), m.size, out.ctypes.data_as(C.POINTER(C.c_short)))
    return out


def remap_nearest(src, fmap) -> np.ndarray:
    s = _img3(src)
    m = _f32(fmap)
    dst = np.empty((m.shape[0], m.shape[1], s.shape[2]), np.float32)
    lib().orc_remap_nearest(_p(s), s.shape[0], s.shape[1], s.shape[2], _p(m), m.shape[0], m.shape[1], _p(dst))
    return dst if np.ndim(src) == 3 else dst[:, :, 0]


def gaussian_kernel(n: int, sigma: float) -> np.ndarray:
    k = np.empty(n, np.float32)
    lib().orc_gaussian_kernel(n, sigma, _p(k))
    return k


def sep_filter(src, k) -> np.ndarray:
    s = _img3(src)
    k = _f32(k)
    tmp = np.empty_like(s)
    dst = np.empty_like(s)
    lib().orc_sep_filter(_p(s), s.shape[0], s.shape[1], s.shape[2], _p(k), k.size, _p(tmp), _p(dst))
    return dst if np.ndim(src) == 3 else dst[:, :, 0]


def resize_nearest(src, drows: int, dcols: int) -> np.ndarray:
    s = _img3(src)
    dst = np.empty((drows, dcols, s.shape[2]), np.float32)
    lib().orc_resize_nearest(_p(s), s.shape[0], s.shape[1], s.shape[2], _p(dst), drows, dcols)
    return dst if np.ndim(src) == 3 else dst[:, :, 0]


def diff_sign(a, b) -> np.ndarray:
    a = _f32(a)
    b = _f32(b)
    dst = np.empty_like(a)
    lib().orc_diff_sign(_p(a), _p(b), a.size, _p(dst))
    return dst


def upscale(src, scale: int) -> np.ndarray:
    s = _img3(src)
    dst = np.empty((s.shape[0] * scale, s.shape[1] * scale, s.shape[2]), np.float32)
    lib().orc_upscale(_p(s), s.shape[0], s.shape[1], s.shape[2], scale, _p(dst))
    return dst if np.ndim(src) == 3 else dst[:, :, 0]


def calc_btv_weights(btv_kernel_size: int, alpha: float) -> np.ndarray:
    """Full btvKernelSize^2 vector as the reference sizes it; only the first
    n_written entries are defined (the rest stay zero)."""
    w = np.zeros(btv_kernel_size * btv_kernel_size + 1, np.float32)
    n = lib().orc_calc_btv_weights(btv_kernel_size, alpha, _p(w))
    return w[:n].copy()


def btv_regularization(src, btv_kernel_size: int, w) -> np.ndarray:
    s = _img3(src)
    w = _f32(w)
    dst = np.empty_like(s)
    lib().orc_btv_regularization(_p(s), s.shape[0], s.shape[1], s.shape[2], btv_kernel_size, _p(w), _p(dst))
    return dst if np.ndim(src) == 3 else dst[:, :, 0]


def add_weighted(a, alpha: float, b, beta: float, gamma: float = 0.0) -> np.ndarray:
    a = _f32(a)
    b = _f32(b)
    dst = np.empty_like(a)
    lib().orc_add_weighted(_p(a), alpha, _p(b), beta, gamma, a.size, _p(dst))
    return dst


def convert_8u(src) -> np.ndarray:
    s = _f32(src)
    dst = np.empty(s.shape, np.uint8)
    lib().orc_convert_8u(_p(s), s.size, dst.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return dst


def btvl1_run(y: Sequence[np.ndarray], rel: Sequence[np.ndarray], base_idx: int, params: BtvParams,
              return_full: bool = False):
    """BTV_L1_Base::run, lib/src/btv_l1.cpp:357-457.  y: K float LR frames, rel: K relative motions."""
    return _solve("orc_btvl1_run", y, rel, base_idx, params, return_full)


def btvl1_process(y: Sequence[np.ndarray], fwd_motions: Sequence[np.ndarray], base_idx: int, params: BtvParams,
                  return_full: bool = False):
    """BTV_L1_Base::process(src, dst, forwardMotions, baseIdx), lib/src/btv_l1.cpp:328-355."""
    return _solve("orc_btvl1_process", y, fwd_motions, base_idx, params, return_full)


def _solve(fn, y, motions, base_idx, params, return_full):
    ys = [_img3(f) for f in y]
    ms = [_f32(m) for m in motions]
    rows, cols, ch = ys[0].shape
    s, B = params.scale, params.btv_kernel_size
    out = np.empty((rows * s - 2 * B, cols * s - 2 * B, ch), np.float32)
    full = np.empty((rows * s, cols * s, ch), np.float32) if return_full else None
    mp = _pp(ms) if ms else (_fp * 1)()
    rc = getattr(lib(), fn)(_pp(ys), mp, len(ys), base_idx, rows, cols, ch, C.byref(params), _p(out),
                            _p(full) if return_full else None)
    if rc != 0:
        raise ValueError("oracle rejected the parameters")
    squeeze = np.ndim(y[0]) == 2
    if squeeze:
        out = out[:, :, 0]
        full = full[:, :, 0] if return_full else None
    return (out, full) if return_full else out


# ---------------------------------------------------------------------------
# quality metrics (oracle/quality_oracle.c)
def psnr_8u(a: np.ndarray, b: np.ndarray) -> float:
    """cv::PSNR of two 8-bit images (rows, cols[, ch]); row pitches are honoured, so views work."""
    L = lib()
    L.orc_psnr_8u.restype = C.c_double
    L.orc_psnr_8u.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]
    assert a.dtype == np.uint8 and b.dtype == np.uint8 and a.shape == b.shape
    ch = a.shape[2] if a.ndim == 3 else 1
    for im in (a, b):  # rows may be strided, pixels within a row are dense
        assert im.strides[1] == ch and (im.ndim == 2 or im.strides[2] == 1)
    return L.orc_psnr_8u(a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0], a.shape[0], a.shape[1], ch)


def mssim(a: np.ndarray, b: np.ndarray):
    """MSSIM of samples/quality_test/main.cpp:140-199.  Returns (value as the reference returns it,
    per-channel means of the SSIM map)."""
    L = lib()
    L.orc_mssim.restype = C.c_double
    L.orc_mssim.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    a3 = np.ascontiguousarray(a if a.ndim == 3 else a[:, :, None], dtype=np.float32)
    b3 = np.ascontiguousarray(b if b.ndim == 3 else b[:, :, None], dtype=np.float32)
    assert a3.shape == b3.shape
    m = (C.c_double * 4)()
    v = L.orc_mssim(a3.ctypes.data_as(_fp), b3.ctypes.data_as(_fp), a3.shape[0], a3.shape[1], a3.shape[2], m)
    return v, np.array(m[:a3.shape[2]])
