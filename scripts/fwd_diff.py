"""Where do the specialised and the generic forward kernels disagree? (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import ops
rows, cols, ch, scale = [int(v) for v in (sys.argv[1:5] or (24, 32, 1, 4))]
amp = float(sys.argv[5]) if len(sys.argv) > 5 else 3.0
rng = np.random.default_rng(7)
H, W = rows * scale, cols * scale
X = (rng.random((H, W, ch)) * 255).astype(np.float32)
yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
rel = np.stack([amp * np.sin(yy / 9 + xx / 13), amp * np.cos(yy / 11 - xx / 7)], -1).astype(np.float32)
gk = ops.gaussian_kernel(5, 0.0)
y = (rng.random((rows, cols, ch)) * 255).astype(np.float32)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
_, bwd_map = ops.build_maps(d(rel), scale)
out = []
for path in (1, 0):
    ops.set_kernel_path(path)
    out.append(ops.forward_sign(d(X), bwd_map, d(y), scale, gk).cpu().numpy())
bad = np.argwhere(out[0] != out[1])
print("mismatches", len(bad), "of", out[0].size)
print(bad[:40].tolist())
