"""Largest size the scale-4 kernels accept (1080p -> 8K): specialised against plain kernels (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import api, ops

rows, cols, ch, K, T = 1080, 1920, 3, 3, 2
g = torch.Generator(device="cuda").manual_seed(5)
y = [(torch.rand((rows, cols, ch), device="cuda", generator=g) * 255).round() for _ in range(K)]
yy, xx = torch.meshgrid(torch.arange(rows, device="cuda", dtype=torch.float32), torch.arange(cols, device="cuda", dtype=torch.float32), indexing="ij")
fl = [torch.stack([1.3 * torch.sin(yy / 90 + k) + 0.7, 0.9 * torch.cos(xx / 70 - k) - 0.4], -1).contiguous() for k in range(K - 1)]
torch.cuda.synchronize()
outs = []
for path in (1, 0):
    ops.set_kernel_path(path)
    sr = api.BTV_L1()
    sr.set("iterations", T)
    out, full = ops.process_device(sr, y, fl, 1, return_full=True)
    outs.append((out.cpu(), full.cpu()))
    del sr
ops.set_kernel_path(0)
same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
print("8K solve:", "identical" if same else "DIFFERENT", tuple(outs[0][0].shape))
sys.exit(0 if same else 1)
