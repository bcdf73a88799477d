"""ctypes binding of include/srx.h (superresolution_amd/libsrx.so).

The library is hand-written HIP for gfx950; there is no CPU fallback.  A
missing library is an error, not a reason to compute somewhere else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRX_LIBRARY selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("SRX_LIBRARY") or os.path.join(_HERE, "libsrx.so")

SRX_OK = 0
SRX_ERR_INVALID, SRX_ERR_STATE, SRX_ERR_DEVICE, SRX_ERR_SOURCE, SRX_ERR_FLOW = 1, 2, 3, 4, 5
SRX_8U, SRX_16U, SRX_32F = 0, 2, 5
SRX_MEM_HOST, SRX_MEM_DEVICE = 0, 1
SRX_FLOW_NONE, SRX_FLOW_TVL1, SRX_FLOW_CALLBACK, SRX_FLOW_FARNEBACK = 0, 1, 2, 3
SRX_EV_READ, SRX_EV_FLOW, SRX_EV_SOLVE, SRX_EV_OUT = 1, 2, 3, 4


class SrxImage(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("rows", C.c_int),
        ("cols", C.c_int),
        ("channels", C.c_int),
        ("depth", C.c_int),
        ("step", C.c_size_t),
        ("mem", C.c_int),
    ]


NEXT_FRAME_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(SrxImage))
RESET_FN = C.CFUNCTYPE(None, C.c_void_p)
FLOW_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(SrxImage), C.POINTER(SrxImage), C.POINTER(SrxImage))
FLOW2_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(SrxImage), C.POINTER(SrxImage), C.POINTER(SrxImage), C.POINTER(SrxImage))

_vp = C.c_void_p
_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_vpp = C.POINTER(C.c_void_p)
_imgp = C.POINTER(SrxImage)

# name -> (restype, argtypes); this table is also what the symbol-export test walks
SIGNATURES = {
    "srx_create_btvl1": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "srx_destroy": (None, [_vp]),
    "srx_last_error": (C.c_char_p, []),
    "srx_set_int": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "srx_set_double": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "srx_get_int": (C.c_int, [_vp, C.c_char_p, _ip]),
    "srx_get_double": (C.c_int, [_vp, C.c_char_p, _dp]),
    "srx_set_flow_builtin": (C.c_int, [_vp, C.c_int]),
    "srx_set_flow_callback": (C.c_int, [_vp, FLOW_FN, _vp]),
    "srx_set_flow_callback2": (C.c_int, [_vp, FLOW2_FN, _vp]),
    "srx_flow_set_int": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "srx_flow_set_double": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "srx_flow_get_int": (C.c_int, [_vp, C.c_char_p, _ip]),
    "srx_flow_get_double": (C.c_int, [_vp, C.c_char_p, _dp]),
    "srx_set_input": (C.c_int, [_vp, NEXT_FRAME_FN, RESET_FN, _vp]),
    "srx_reset": (C.c_int, [_vp]),
    "srx_next_frame": (C.c_int, [_vp, _imgp]),
    "srx_set_output_range": (C.c_int, [_vp, C.c_int, C.c_int]),
    "srx_process": (C.c_int, [_vp, _imgp, C.c_int, _imgp, C.c_int, _imgp]),
    "srx_flow_calc": (C.c_int, [_vp, _imgp, _imgp, _imgp, _imgp]),
    "srx_collect_garbage": (C.c_int, [_vp]),
    "srx_flow_calc_device": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, C.POINTER(C.c_longlong)]),
    "srx_flow_calc_device_pairs": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, C.POINTER(C.c_longlong),
                                             C.c_int]),
    "srx_flow_pixel_iterations": (C.c_int, [_vp, C.POINTER(C.c_longlong)]),
    "srx_drain_events": (C.c_int, [_vp, _ip, C.c_int]),
    "srx_ring_slot": (C.c_int, [C.c_int, C.c_int]),
    "srx_window": (None, [C.c_int, C.c_int, C.c_int, _ip, _ip, _ip]),
    "srx_calc_btv_weights": (C.c_int, [C.c_int, C.c_double, _fp]),
    "srx_gaussian_kernel": (C.c_int, [C.c_int, C.c_double, _fp]),
    "srx_span_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, _ip]),
    "srx_k_relative_motions": (C.c_int, [_vpp, _vpp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "srx_k_build_maps": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "srx_k_resize_cubic": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp]),
    "srx_k_btv_regularization": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _vp, _vp]),
    "srx_k_forward_sign": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _fp, C.c_int, _vp, _vp]),
    "srx_k_backproject_update": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vpp, _vpp, C.c_int, C.c_int, _fp,
                                           C.c_int, C.c_int, _fp, C.c_double, C.c_double, _vp]),
    "srx_k_crop_convert": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "srx_run": (C.c_int, [_vp, _vpp, _vpp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "srx_process_device": (C.c_int, [_vp, _vpp, _vpp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "srx_quality_psnr": (C.c_int, [C.POINTER(SrxImage), C.POINTER(SrxImage), C.POINTER(C.c_double), _vp]),
    "srx_quality_psnr_noalloc": (C.c_int, [C.POINTER(SrxImage), C.POINTER(SrxImage), C.POINTER(C.c_double), _vp]),
    "srx_quality_mssim": (C.c_int, [C.POINTER(SrxImage), C.POINTER(SrxImage), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), _vp]),
    "srx_set_kernel_path": (C.c_int, [C.c_int]),
    "srx_get_kernel_path": (C.c_int, []),
    "srx_kernel_times": (C.c_int, [_vp, _dp, _dp, C.POINTER(C.c_long)]),
    "srx_k_copy": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "srx_stream": (_vp, [_vp]),
    "srx_synchronize": (C.c_int, [_vp]),
}

_lib = None


class SrxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"srx error {code}: {msg}")
        self.code = code


def lib() -> C.CDLL:
    """Load libsrx.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP library has not been built and there is no fallback path")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != SRX_OK:
        raise SrxError(rc, lib().srx_last_error().decode())


def ptr_array(ptrs):
    """Host array of (device) pointers for the *_host pointer-list arguments."""
    arr = (C.c_void_p * max(len(ptrs), 1))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr
