"""Seeded synthetic sequences for tests and benchmarks.

A band-limited texture (random-phase sinusoids) plus a few hard-edged shapes
is observed through a slowly moving camera (sub-pixel translation, slight
rotation and zoom per frame), so the true inter-frame motion is known in
closed form.  Ground-truth frames are degraded the way the reference's quality
harness does it (samples/quality_test/main.cpp:127-138): 5x5 Gaussian blur
with sigma 0, i.e. {1,4,6,4,1}/16 with BORDER_REFLECT_101, then nearest
decimation anchored at (y*scale, x*scale).  Noise (sigma-10 Gaussian, 1/500
spikes) is optional and comes from torch's seeded generator, since OpenCV's
theRNG stream cannot be reproduced.

torch is used as the array library so that the same code fills host arrays in
tests (device="cpu") and HBM-resident inputs in the benchmark (device="cuda").
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

import torch


@dataclass
class Camera:
    """Per-frame similarity transform, texture coords = A @ (x, y) + b in HR pixels."""

    a00: float
    a01: float
    a10: float
    a11: float
    b0: float
    b1: float


def camera_path(n_frames: int, seed: int = 1234, max_shift: float = 1.7, max_rot: float = 0.002,
                max_zoom: float = 0.002, rot_growth: bool = False) -> List[Camera]:
    """Random-walk translation (steps of at most max_shift HR pixels per axis) with a small rotation and zoom
    per frame.  The rotation is stationary: every frame's angle is uniform in +-max_rot.  rot_growth=True lets
    the amplitude grow with sqrt(frame index), the model the committed golden files were generated with."""
    g = torch.Generator().manual_seed(seed)
    cams = []
    tx = ty = 0.0
    for t in range(n_frames):
        r = torch.rand(4, generator=g).tolist()
        tx += (r[0] * 2 - 1) * max_shift
        ty += (r[1] * 2 - 1) * max_shift
        ang = (r[2] * 2 - 1) * max_rot * ((t + 1) ** 0.5 if rot_growth else 1.0)
        z = 1.0 + (r[3] * 2 - 1) * max_zoom
        c, s = math.cos(ang) * z, math.sin(ang) * z
        cams.append(Camera(c, -s, s, c, tx, ty))
    return cams


def _texture(u: torch.Tensor, v: torch.Tensor, channels: int, seed: int) -> torch.Tensor:
    """Texture value in [0, 255] at texture coordinates (u, v); returns (..., channels)."""
    g = torch.Generator().manual_seed(seed + 77)
    n_waves = 10
    out = []
    for _c in range(channels):
        p = torch.rand(n_waves, 4, generator=g)
        acc = torch.full_like(u, 128.0)
        for i in range(n_waves):
            wl = 6.0 + 90.0 * float(p[i, 0]) ** 2  # wavelength in HR pixels
            th = 2 * math.pi * float(p[i, 1])
            ph = 2 * math.pi * float(p[i, 2])
            amp = 10.0 + 14.0 * float(p[i, 3])
            k = 2 * math.pi / wl
            acc = acc + amp * torch.sin(k * math.cos(th) * u + k * math.sin(th) * v + ph)
        out.append(acc)
    img = torch.stack(out, dim=-1)
    # hard-edged shapes shared by all channels: checker patches and discs on a coarse lattice
    cell = 97.0
    iu = torch.floor(u / cell)
    iv = torch.floor(v / cell)
    fu = u / cell - iu
    fv = v / cell - iv
    sel = torch.remainder(iu * 3 + iv * 5, 7)
    box = ((fu > 0.25) & (fu < 0.6) & (fv > 0.3) & (fv < 0.7) & (sel == 1)).to(img.dtype)
    disc = ((((fu - 0.5) ** 2 + (fv - 0.5) ** 2) < 0.09) & (sel == 4)).to(img.dtype)
    bars = ((torch.remainder(torch.floor(fu * 16), 2) == 0) & (fv > 0.2) & (fv < 0.5) & (sel == 6)).to(img.dtype)
    img = img + 60.0 * box[..., None] - 70.0 * disc[..., None] + 50.0 * bars[..., None]
    return img.clamp_(0.0, 255.0)


def ground_truth_frame(cam: Camera, hr_rows: int, hr_cols: int, channels: int, seed: int, device="cpu") -> torch.Tensor:
    """8-bit ground-truth HR frame (rows, cols, channels) uint8."""
    y = torch.arange(hr_rows, device=device, dtype=torch.float32)[:, None]
    x = torch.arange(hr_cols, device=device, dtype=torch.float32)[None, :]
    u = cam.a00 * x + cam.a01 * y + cam.b0
    v = cam.a10 * x + cam.a11 * y + cam.b1
    img = _texture(u, v, channels, seed)
    return torch.round(img).to(torch.uint8)


def _reflect101_index(n: int, pad: int, device) -> torch.Tensor:
    idx = torch.arange(-pad, n + pad, device=device)
    idx = idx.abs()
    idx = torch.where(idx >= n, 2 * (n - 1) - idx, idx)
    return idx


def degrade(frame_u8: torch.Tensor, scale: int, noise_sigma: float = 0.0, spike_freq: int = 0,
            gen: torch.Generator | None = None) -> torch.Tensor:
    """Blur {1,4,6,4,1}/16 (REFLECT_101), nearest decimation at (y*s, x*s), optional noise; uint8 out."""
    f = frame_u8.to(torch.float32)
    rows, cols = f.shape[0], f.shape[1]
    k = (0.0625, 0.25, 0.375, 0.25, 0.0625)
    ix = _reflect101_index(cols, 2, f.device)
    iy = _reflect101_index(rows, 2, f.device)
    fp = f[:, ix]
    h = sum(k[j] * fp[:, j:j + cols] for j in range(5))
    hp = h[iy]
    b = sum(k[j] * hp[j:j + rows] for j in range(5))
    d = b[::scale, ::scale]
    if noise_sigma > 0:
        d = d + noise_sigma * torch.randn(d.shape, generator=gen, device="cpu").to(d.device)
    d = torch.round(d).clamp_(0, 255).to(torch.uint8)
    if spike_freq > 0:
        mask = torch.randint(0, spike_freq, d.shape[:2], generator=gen, device="cpu").to(d.device) < 1
        d[mask] = 255
    return d.contiguous()


def true_forward_flow(cam0: Camera, cam1: Camera, lr_rows: int, lr_cols: int, scale: int, device="cpu") -> torch.Tensor:
    """Exact flow frame0 -> frame1 in LR pixel units, (rows, cols, 2) float32 with (dx, dy) interleaved.

    Content seen at HR pixel p in frame 0 sits at texture coords A0 p + b0 and is seen in frame 1 at
    q = A1^-1 (A0 p + b0 - b1).  LR pixel (x, y) samples HR pixel (s*x, s*y), so the LR flow is (q - p) / s.
    """
    y = torch.arange(lr_rows, device=device, dtype=torch.float64)[:, None] * scale
    x = torch.arange(lr_cols, device=device, dtype=torch.float64)[None, :] * scale
    u = cam0.a00 * x + cam0.a01 * y + cam0.b0 - cam1.b0
    v = cam0.a10 * x + cam0.a11 * y + cam0.b1 - cam1.b1
    det = cam1.a00 * cam1.a11 - cam1.a01 * cam1.a10
    qx = (cam1.a11 * u - cam1.a01 * v) / det
    qy = (-cam1.a10 * u + cam1.a00 * v) / det
    flow = torch.stack(((qx - x) / scale, (qy - y) / scale), dim=-1)
    return flow.to(torch.float32).contiguous()


def make_sequence(n_frames: int, lr_rows: int, lr_cols: int, channels: int, scale: int, seed: int = 1234,
                  device="cpu", noise: bool = False, keep_gold: bool = False, max_shift: float = 1.7,
                  max_rot: float = 0.002,
                  rot_growth: bool = False) -> Tuple[List[torch.Tensor], List[torch.Tensor], List[torch.Tensor]]:
    """Returns (lr_frames uint8, true forward flows (n-1), gold HR frames uint8 or []).

    max_shift bounds the camera translation per frame (LR pixels), max_rot scales its rotation (radians)."""
    cams = camera_path(n_frames, seed, max_shift=max_shift, max_rot=max_rot, rot_growth=rot_growth)
    gen = torch.Generator().manual_seed(seed + 1)
    lr, gold = [], []
    for t in range(n_frames):
        hrf = ground_truth_frame(cams[t], lr_rows * scale, lr_cols * scale, channels, seed, device)
        lr.append(degrade(hrf, scale, 10.0 if noise else 0.0, 500 if noise else 0, gen))
        if keep_gold:
            gold.append(hrf)
    flows = [true_forward_flow(cams[t], cams[t + 1], lr_rows, lr_cols, scale, device) for t in range(n_frames - 1)]
    return lr, flows, gold
