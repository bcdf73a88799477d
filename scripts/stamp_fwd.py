"""Per-phase cycle totals of the forward kernel's waves from a -DSRX_STAMPS build (development aid).
The update kernel of the same iteration overwrites the stamps, so the solve stops after the forward
launch of iteration 1 by way of the stage-level entry point.
usage: SRX_LIBRARY=variants/libsrx_stamps.so python scripts/stamp_fwd.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from superresolution_amd import api, ops, synth, _lib

rows, cols, ch, scale, K = 540, 960, 3, 4, 9
lr, flows, _ = synth.make_sequence(K, rows, cols, ch, scale, seed=1234, device="cuda", max_shift=1.7, max_rot=0.002)
y = [f.to(torch.float32) for f in lr]
if "--tvl1" in sys.argv:  # motions estimated by the built-in TV-L1 flow instead of the exact ones
    est = api.BTV_L1()
    est.setOpticalFlow(api.DualTVL1())
    flows = [torch.from_numpy(est.calcFlow(lr[i].cpu().numpy(), lr[i + 1].cpu().numpy())).cuda() for i in range(K - 1)]
sr = api.BTV_L1()
sr.set("iterations", 3)
sr.set("lambda", 0.0)  # the update kernel then writes stamps 0, 5, 6 only... not slots 1-4
ops.process_device(sr, y, flows, K // 2)
torch.cuda.synchronize()
lib = _lib.lib()
n_waves = 256 * 9
buf = np.zeros(n_waves * 8, dtype=np.uint64)
rc = lib.srx_debug_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
assert rc == 0, rc
t = buf.reshape(n_waves, 8).astype(np.int64)
names = ["prologue (first window, first row pass)", "compute (issue, gathers, passes)",
         "put (wait for staged rows, LDS writes)", "barrier", "whole wave"]
print(f"{n_waves} waves; ticks summed over the tile's rows: mean / median / p90")
for i, nm in enumerate(names):
    print(f"  {nm:44s} {t[:, i].mean():9.0f} {np.median(t[:, i]):9.0f} {np.percentile(t[:, i], 90):9.0f}")
print("by wave of the block (mean ticks): wave, prologue, compute, put, barrier, whole")
bw = t.reshape(256, 9, 8)
for w in range(9):
    print("  ", w, " ".join(f"{bw[:, w, i].mean():9.0f}" for i in range(5)))
whole = bw[:, :, 4].max(axis=1)
print("slowest wave per block: mean", whole.mean(), "min", whole.min(), "max", whole.max())
kinds = np.stack([(bw[:, :, 7] >> (16 * i)) & 0xFFFF for i in range(4)], axis=-1)
print("rows by kind (mixed, inside, outside, identity), summed over blocks, by wave:")
for w in range(9):
    print("  ", w, kinds[:, w, :].sum(axis=0))
