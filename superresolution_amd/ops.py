"""Stage-level entry points of libsrx.so on torch CUDA tensors.

torch only supplies device memory and the current stream; every function here
calls a srx_k_* / srx_run C-ABI symbol and raises if the library is missing.
Layouts: images (rows, cols, ch) float32; index maps (rows, cols, 2) int16
(x, y); signs (rows, cols) int32 words holding sign+1 of channel c in bits
10c..10c+1.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np
import torch

from . import _lib as L


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, dtype, what: str):
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError(f"{what}: expected a contiguous CUDA tensor of {dtype}")
    return t


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def calc_btv_weights(btv_kernel_size: int, alpha: float) -> np.ndarray:
    w = np.zeros(btv_kernel_size * btv_kernel_size + 1, np.float32)
    n = L.lib().srx_calc_btv_weights(btv_kernel_size, alpha, _fptr(w))
    return w[:n].copy()


def gaussian_kernel(n: int, sigma: float) -> np.ndarray:
    k = np.zeros(n, np.float32)
    L.check(L.lib().srx_gaussian_kernel(n, sigma, _fptr(k)))
    return k


def relative_motions(fwd: Sequence[torch.Tensor], base_idx: int, rows: int, cols: int):
    n = len(fwd) + 1
    dev = fwd[0].device if fwd else "cuda"
    for f in fwd:
        _chk(f, torch.float32, "forward motion")
    rel = [torch.empty((rows, cols, 2), dtype=torch.float32, device=dev) for _ in range(n)]
    L.check(L.lib().srx_k_relative_motions(L.ptr_array([f.data_ptr() for f in fwd]),
                                           L.ptr_array([r.data_ptr() for r in rel]), n, base_idx, rows, cols,
                                           _stream()))
    return rel


def build_maps(rel: torch.Tensor, scale: int):
    _chk(rel, torch.float32, "relative motion")
    rows, cols = rel.shape[0], rel.shape[1]
    fwd = torch.empty((rows * scale, cols * scale, 2), dtype=torch.int16, device=rel.device)
    bwd = torch.empty_like(fwd)
    L.check(L.lib().srx_k_build_maps(rel.data_ptr(), rows, cols, scale, fwd.data_ptr(), bwd.data_ptr(), _stream()))
    return fwd, bwd


def resize_cubic(src: torch.Tensor, drows: int, dcols: int) -> torch.Tensor:
    _chk(src, torch.float32, "src")
    s3 = src if src.dim() == 3 else src[:, :, None]
    dst = torch.empty((drows, dcols, s3.shape[2]), dtype=torch.float32, device=src.device)
    L.check(L.lib().srx_k_resize_cubic(s3.data_ptr(), s3.shape[0], s3.shape[1], s3.shape[2], dst.data_ptr(), drows,
                                       dcols, _stream()))
    return dst if src.dim() == 3 else dst[:, :, 0]


def btv_regularization(src: torch.Tensor, btv_kernel_size: int, w: np.ndarray) -> torch.Tensor:
    _chk(src, torch.float32, "src")
    s3 = src if src.dim() == 3 else src[:, :, None]
    dst = torch.empty_like(s3)
    w = np.ascontiguousarray(w, np.float32)
    L.check(L.lib().srx_k_btv_regularization(s3.data_ptr(), s3.shape[0], s3.shape[1], s3.shape[2], btv_kernel_size,
                                             _fptr(w), dst.data_ptr(), _stream()))
    return dst if src.dim() == 3 else dst[:, :, 0]


def forward_sign(X: torch.Tensor, bwd_map: torch.Tensor, y: torch.Tensor, scale: int, gauss: np.ndarray):
    _chk(X, torch.float32, "X")
    _chk(bwd_map, torch.int16, "bwd_map")
    _chk(y, torch.float32, "y")
    X3 = X if X.dim() == 3 else X[:, :, None]
    H, W, ch = X3.shape
    sign = torch.zeros((H // scale, W // scale), dtype=torch.int32, device=X.device)
    g = np.ascontiguousarray(gauss, np.float32)
    L.check(L.lib().srx_k_forward_sign(X3.data_ptr(), H, W, ch, bwd_map.data_ptr(), y.data_ptr(), scale, _fptr(g),
                                       g.size, sign.data_ptr(), _stream()))
    return sign


def backproject_update(X: torch.Tensor, fwd_maps: Sequence[torch.Tensor], signs: Sequence[torch.Tensor], scale: int,
                       gauss: np.ndarray, btv_kernel_size: int, w: np.ndarray, tau: float, lam: float):
    _chk(X, torch.float32, "X")
    X3 = X if X.dim() == 3 else X[:, :, None]
    H, W, ch = X3.shape
    for m in fwd_maps:
        _chk(m, torch.int16, "fwd_map")
    for s in signs:
        _chk(s, torch.int32, "sign")
    out = torch.empty_like(X3)
    g = np.ascontiguousarray(gauss, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    L.check(L.lib().srx_k_backproject_update(X3.data_ptr(), out.data_ptr(), H, W, ch,
                                             L.ptr_array([m.data_ptr() for m in fwd_maps]),
                                             L.ptr_array([s.data_ptr() for s in signs]), len(fwd_maps), scale,
                                             _fptr(g), g.size, btv_kernel_size, _fptr(w), tau, lam, _stream()))
    return out if X.dim() == 3 else out[:, :, 0]


def crop_convert(X: torch.Tensor, border: int, to_8u: bool) -> torch.Tensor:
    _chk(X, torch.float32, "X")
    X3 = X if X.dim() == 3 else X[:, :, None]
    H, W, ch = X3.shape
    out = torch.empty((H - 2 * border, W - 2 * border, ch), dtype=torch.uint8 if to_8u else torch.float32,
                      device=X.device)
    L.check(L.lib().srx_k_crop_convert(X3.data_ptr(), H, W, ch, border, out.data_ptr(),
                                       L.SRX_8U if to_8u else L.SRX_32F, _stream()))
    return out if X.dim() == 3 else out[:, :, 0]


def pack_signs(s: np.ndarray) -> np.ndarray:
    """(rows, cols, ch) array of -1/0/1 -> the packed int32 sign words."""
    s = np.asarray(s)
    if s.ndim == 2:
        s = s[:, :, None]
    out = np.zeros(s.shape[:2], np.int32)
    for c in range(s.shape[2]):
        out |= (s[:, :, c].astype(np.int32) + 1) << (10 * c)
    return out


def unpack_signs(words, ch: int) -> np.ndarray:
    """Packed int32 sign words -> (rows, cols, ch) int8 array of -1/0/1."""
    wd = np.asarray(words).astype(np.int64)
    return np.stack([((wd >> (10 * c)) & 3) - 1 for c in range(ch)], axis=-1).astype(np.int8)


def set_kernel_path(path: int) -> None:
    """0: specialised kernels where the parameters allow (default); 1: generic kernels only."""
    L.check(L.lib().srx_set_kernel_path(path))


def _solve(fn_name: str, sr, y: Sequence[torch.Tensor], motions: Sequence[torch.Tensor], base_idx: int,
           return_full: bool, wait: bool = True):
    for t in y:
        _chk(t, torch.float32, "frame")
    for t in motions:
        _chk(t, torch.float32, "motion")
    y3 = [t if t.dim() == 3 else t[:, :, None] for t in y]
    rows, cols, ch = y3[0].shape
    s, B = sr.getInt("scale"), sr.getInt("btvKernelSize")
    out = torch.empty((rows * s - 2 * B, cols * s - 2 * B, ch), dtype=torch.float32, device=y[0].device)
    full = torch.empty((rows * s, cols * s, ch), dtype=torch.float32, device=y[0].device) if return_full else None
    fn = getattr(L.lib(), fn_name)
    if wait:
        torch.cuda.synchronize()  # inputs were produced on torch's stream, the solve runs on the handle's
    L.check(fn(sr.handle, L.ptr_array([t.data_ptr() for t in y3]), L.ptr_array([t.data_ptr() for t in motions]),
               len(y3), base_idx, rows, cols, ch, out.data_ptr(), full.data_ptr() if return_full else None))
    if wait:
        sr.synchronize()
    if y[0].dim() == 2:
        out = out[:, :, 0]
        full = full[:, :, 0] if return_full else None
    return (out, full) if return_full else out


def run(sr, y, rel, base_idx: int, return_full: bool = False):
    """BTV_L1_Base::run (lib/src/btv_l1.cpp:357-457) on device tensors with sr's parameters."""
    return _solve("srx_run", sr, y, rel, base_idx, return_full)


def process_device(sr, y, fwd_motions, base_idx: int, return_full: bool = False, wait: bool = True):
    """BTV_L1_Base::process with forward motions (lib/src/btv_l1.cpp:328-355) on device tensors.

    wait=False neither waits for torch's stream before the solve nor for the solve itself: the caller
    makes sure the inputs are ready and calls sr.synchronize() before reading the result.
    """
    return _solve("srx_process_device", sr, y, fwd_motions, base_idx, return_full, wait)
