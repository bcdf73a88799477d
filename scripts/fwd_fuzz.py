"""Random shapes and motions: specialised forward kernel against the generic one (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
gk = ops.gaussian_kernel(5, 0.0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bad = 0
for it in range(n):
    rows, cols = int(rng.integers(1, 90)), int(rng.integers(1, 200))
    ch = int(rng.choice([1, 3]))
    H, W = rows * 4, cols * 4
    X = (rng.random((H, W, ch)) * 255).astype(np.float32)
    kind = int(rng.integers(0, 4))
    amp = float(rng.choice([0.2, 1.0, 3.0, 8.0]))
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    rel = np.stack([amp * np.sin(yy / 9 + xx / 13 + it), amp * np.cos(yy / 11 - xx / 7 + it)], -1).astype(np.float32)
    if kind == 1:
        rel += rng.standard_normal(rel.shape).astype(np.float32) * 2
    elif kind == 2:
        rel += np.float32(rng.uniform(-6, 6))
    elif kind == 3:
        rel *= rng.integers(0, 2, rel.shape).astype(np.float32)
    y = (rng.random((rows, cols, ch)) * 255).astype(np.float32)
    _, bwd_map = ops.build_maps(d(rel), 4)
    out = []
    for path in (1, 0):
        ops.set_kernel_path(path)
        out.append(ops.forward_sign(d(X), bwd_map, d(y), 4, gk).cpu().numpy())
    m = int((out[0] != out[1]).sum())
    if m:
        bad += 1
        print("MISMATCH", rows, cols, ch, kind, amp, m, flush=True)
print("cases", n, "bad", bad)
sys.exit(1 if bad else 0)
