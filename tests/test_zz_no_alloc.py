"""srx_quality_psnr_noalloc (the entry point that makes no device allocation: partial sums in a static array in
every GPU's memory) against srx_quality_psnr, through the C ABI, bit for bit: two calls in a row, the second on a
smaller pair so that what the first call left in the array would show if it were read; regions of interest with a
row pitch; the size limit and the host-image refusal."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(x: float) -> bytes:
    return struct.pack("<d", x)


def _pair(torch, rows, cols, ch, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randint(0, 256, (rows, cols, ch), dtype=torch.uint8, device="cuda", generator=g)
    noise = torch.randint(-9, 10, (rows, cols, ch), dtype=torch.int16, device="cuda", generator=g)
    b = (a.to(torch.int16) + noise).clamp(0, 255).to(torch.uint8)
    return a, b


def test_no_allocation_psnr_matches_on_two_calls_in_a_row_the_second_smaller():
    import torch
    from superresolution_amd import quality

    big = _pair(torch, 2146, 3826, 3, 1)      # the delivered frame of the headline configuration
    small = _pair(torch, 37, 53, 3, 2)        # one segment per row, ragged tail
    want_big, want_small = quality.PSNR(*big), quality.PSNR(*small)
    got_big = quality.PSNR(*big, allocate=False)
    got_small = quality.PSNR(*small, allocate=False)
    assert _bits(got_big) == _bits(want_big)
    assert _bits(got_small) == _bits(want_small)
    # and once more the other way round, after the array has held both
    assert _bits(quality.PSNR(*big, allocate=False)) == _bits(want_big)
    # equal images: the total must be written as zero, not left over from the call before
    assert _bits(quality.PSNR(small[0], small[0], allocate=False)) == _bits(quality.PSNR(small[0], small[0]))
    # against the oracle as well
    import oracle

    assert abs(got_small - oracle.psnr_8u(small[0].cpu().numpy(), small[1].cpu().numpy())) < 1e-9


def test_no_allocation_psnr_on_regions_with_a_row_pitch():
    import torch
    from superresolution_amd import quality

    a, b = _pair(torch, 300, 5000, 1, 3)      # rows longer than one 4096-byte segment
    ra, rb = a[7:-11, 13:-5], b[7:-11, 13:-5]
    assert _bits(quality.PSNR(ra, rb, allocate=False)) == _bits(quality.PSNR(ra, rb))


def test_no_allocation_psnr_refuses_what_it_cannot_hold():
    import torch
    from superresolution_amd import quality, _lib as L

    a = torch.zeros((65537, 8, 1), dtype=torch.uint8, device="cuda")
    with pytest.raises(L.SrxError, match="limit is 65536"):
        quality.PSNR(a, a, allocate=False)
    assert quality.PSNR(a[:65536], a[:65536], allocate=False) == quality.PSNR(a[:65536], a[:65536])
    h = np.zeros((8, 8), np.uint8)
    with pytest.raises(L.SrxError, match="device memory"):
        quality.PSNR(h, h, allocate=False)


def test_no_allocation_psnr_allocates_nothing_and_checks_who_owns_the_images():
    import ctypes as C

    import torch
    from superresolution_amd import quality, _lib as L

    a, b = _pair(torch, 480, 640, 3, 5)
    want = quality.PSNR(a, b)
    quality.PSNR(a, b, allocate=False)        # the first call on a device looks the array's address up
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        assert _bits(quality.PSNR(a, b, allocate=False)) == _bits(want)
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after == free_before          # hipMemGetInfo: not a byte of device memory taken
    # a host pointer labelled as device memory is refused before any kernel sees it
    host = torch.zeros((16, 16, 1), dtype=torch.uint8).pin_memory()
    keep: list = []
    hh = quality._header(host, keep)
    hh.mem = L.SRX_MEM_DEVICE
    out = C.c_double()
    rc = L.lib().srx_quality_psnr_noalloc(C.byref(hh), C.byref(hh), C.byref(out), None)
    assert rc == L.SRX_ERR_INVALID
    with pytest.raises(L.SrxError, match="not in device memory"):
        L.check(rc)
    # images of another GPU than the current one are refused (needs two visible GPUs)
    if torch.cuda.device_count() > 1:
        with torch.cuda.device(1):
            a1, b1 = a.to("cuda:1"), b.to("cuda:1")
            assert _bits(quality.PSNR(a1, b1, allocate=False)) == _bits(want)   # device 1 current: its own array
        with pytest.raises(L.SrxError, match="current device"):
            ha, hb = quality._header(a1, keep), quality._header(b1, keep)
            L.check(L.lib().srx_quality_psnr_noalloc(C.byref(ha), C.byref(hb), C.byref(out), None))
