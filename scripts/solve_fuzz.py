"""Random solver configurations: specialised / tiled kernels against the plain per-pixel kernels of the
library (kernel path 1), whole solves compared bit for bit (development aid; no CPU code involved)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import api, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bad = 0
for it in range(n):
    scale = int(rng.choice([2, 3, 4, 4, 4]))
    big = rng.random() < 0.25
    rows = int(rng.integers(60, 200)) if big else int(rng.integers(3, 60))
    cols = int(rng.integers(100, 400)) if big else int(rng.integers(3, 90))
    ch = int(rng.choice([1, 3]))
    K = int(rng.integers(1, 12))
    base = int(rng.integers(0, K))
    T = int(rng.integers(1, 5))
    blur = int(rng.choice([5, 5, 5, 3, 7, 1]))
    sigma = float(rng.choice([0.0, 0.0, 1.1]))
    btv = int(rng.choice([7, 7, 5, 3, 9]))
    if min(rows, cols) * scale <= 2 * btv:
        continue
    lam = float(rng.choice([0.03, 0.0, 0.1]))
    amp = float(rng.choice([0.1, 0.4, 1.5, 4.0]))
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    fl = []
    for k in range(K - 1):
        f = np.stack([amp * np.sin(yy / 9 + xx / 13 + it + k), amp * np.cos(yy / 11 - xx / 7 + it - k)], -1)
        f = f + rng.uniform(-amp, amp, 2)
        if rng.random() < 0.3:
            f = f + rng.standard_normal(f.shape) * 0.5
        fl.append(d(f.astype(np.float32)))
    y = [d((rng.random((rows, cols, ch)) * 255).round().astype(np.float32)) for _ in range(K)]
    outs = []
    for path in (1, 0):
        ops.set_kernel_path(path)
        sr = api.BTV_L1()
        for name, v in (("scale", scale), ("iterations", T), ("blurKernelSize", blur), ("blurSigma", sigma),
                        ("btvKernelSize", btv), ("lambda", lam)):
            sr.set(name, v)
        out, full = ops.process_device(sr, y, fl, base, return_full=True)
        outs.append((out.cpu().numpy(), full.cpu().numpy()))
    ops.set_kernel_path(0)
    m = int((outs[0][1] != outs[1][1]).sum()) + int((outs[0][0] != outs[1][0]).sum())
    if m:
        bad += 1
        print("MISMATCH", dict(scale=scale, rows=rows, cols=cols, ch=ch, K=K, base=base, T=T, blur=blur, sigma=sigma,
                               btv=btv, lam=lam, amp=amp), m, flush=True)
    elif it % 10 == 0:
        print("ok", it, scale, rows, cols, ch, K, flush=True)
print("cases", n, "bad", bad)
sys.exit(1 if bad else 0)
