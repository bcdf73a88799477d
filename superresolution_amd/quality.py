"""Quality metrics of the reference's quality harness (samples/quality_test/main.cpp) on the GPU.

PSNR and MSSIM carry the names they have in the harness; both go through the C ABI
(srx_quality_psnr / srx_quality_mssim) and raise if the HIP library is missing.
Images are numpy arrays or torch tensors, (rows, cols) or (rows, cols, ch), uint8 or
float32.  Views whose rows are strided (a region of interest such as
goldFrame(inner), main.cpp:320) are passed without a copy.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _header(a, keep: list) -> L.SrxImage:
    im = L.SrxImage()
    is_torch = type(a).__module__.startswith("torch")
    if is_torch:
        import torch

        if a.dtype not in (torch.uint8, torch.float32):
            raise TypeError("images must be uint8 or float32")
        shape, strides, esz = tuple(a.shape), tuple(s * a.element_size() for s in a.stride()), a.element_size()
        depth = L.SRX_8U if a.dtype == torch.uint8 else L.SRX_32F
        mem = L.SRX_MEM_DEVICE if a.is_cuda else L.SRX_MEM_HOST
    else:
        a = np.asarray(a)
        if a.dtype not in (np.uint8, np.float32):
            raise TypeError("images must be uint8 or float32")
        shape, strides, esz = a.shape, a.strides, a.itemsize
        depth = L.SRX_8U if a.dtype == np.uint8 else L.SRX_32F
        mem = L.SRX_MEM_HOST
    if len(shape) == 2:
        shape, strides = shape + (1,), strides + (esz,)
    if len(shape) != 3:
        raise ValueError("images must be (rows, cols) or (rows, cols, ch)")
    dense_inner = strides[2] == esz and strides[1] == esz * shape[2] and strides[0] >= esz * shape[2] * shape[1]
    if not dense_inner:  # anything but a row-strided view is made dense
        a = a.contiguous() if is_torch else np.ascontiguousarray(a)
        strides = (esz * shape[2] * shape[1], esz * shape[2], esz)
    keep.append(a)
    im.data = a.data_ptr() if is_torch else a.ctypes.data
    im.rows, im.cols, im.channels = shape
    im.depth, im.step, im.mem = depth, strides[0], mem
    return im


def _stream(a):
    if type(a).__module__.startswith("torch") and a.is_cuda:
        import torch

        return C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
    return None


def PSNR(a, b, allocate: bool = True) -> float:
    """cv::PSNR of two 8-bit images (samples/quality_test/main.cpp:320,323).  allocate=False takes the entry point
    that makes no device allocation (device images of bounded size only, see srx_quality_psnr_noalloc)."""
    keep: list = []
    ha, hb = _header(a, keep), _header(b, keep)
    out = C.c_double()
    fn = L.lib().srx_quality_psnr if allocate else L.lib().srx_quality_psnr_noalloc
    L.check(fn(C.byref(ha), C.byref(hb), C.byref(out), _stream(a)))
    return out.value


def MSSIM(a, b, as_reference: bool = True) -> float:
    """MSSIM() of samples/quality_test/main.cpp:140-199.

    as_reference=True returns what the reference's function returns, which for 3-channel images is
    (mean[0] + mean[1] + mean[3]) / 3 with mean[3] = 0 (main.cpp:198 reads component 3 where 2 is
    meant); as_reference=False returns the mean over all channels.
    """
    m, ref = mssim_channels(a, b)
    return ref if as_reference else float(np.mean(m))


def mssim_channels(a, b):
    """(per-channel means of the SSIM map, value as the reference returns it)."""
    keep: list = []
    ha, hb = _header(a, keep), _header(b, keep)
    per = (C.c_double * 4)()
    ref = C.c_double()
    L.check(L.lib().srx_quality_mssim(C.byref(ha), C.byref(hb), per, C.byref(ref), _stream(a)))
    return np.array(per[:ha.channels]), ref.value
