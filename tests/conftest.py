import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def rng():
    return np.random.default_rng(20121212)


def smooth_motion(rng, rows, cols, amp=1.5, seed_scale=8):
    """Random smooth LR motion field (rows, cols, 2) float32: coarse noise, bilinearly upsampled."""
    gy, gx = rows // seed_scale + 2, cols // seed_scale + 2
    coarse = rng.standard_normal((gy, gx, 2)).astype(np.float32) * amp
    ys = np.linspace(0, gy - 1.001, rows, dtype=np.float32)
    xs = np.linspace(0, gx - 1.001, cols, dtype=np.float32)
    y0 = np.floor(ys).astype(int)
    x0 = np.floor(xs).astype(int)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    a = coarse[y0][:, x0]
    b = coarse[y0][:, x0 + 1]
    c = coarse[y0 + 1][:, x0]
    d = coarse[y0 + 1][:, x0 + 1]
    return np.ascontiguousarray((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy, dtype=np.float32)
