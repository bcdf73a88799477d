"""Per-phase cycle breakdown of the update kernel from a -DSRX_STAMPS build (development aid).
usage: SRX_LIBRARY=variants/libsrx_stamps.so python scripts/stamp_stats.py"""
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
sr = api.BTV_L1()
sr.set("iterations", 6)
ops.process_device(sr, y, flows, K // 2)
torch.cuda.synchronize()
lib = _lib.lib()
n_waves = 32416
buf = np.zeros(n_waves * 8, dtype=np.uint64)
rc = lib.srx_debug_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
assert rc == 0, rc
t = buf.reshape(n_waves, 8).astype(np.int64)
names = ["start->tile0 staged", "BTV ch0", "BTV ch1", "BTV ch2", "frame loop", "update+store"]
d = np.diff(t[:, :7], axis=1)
ok = (d >= 0).all(axis=1) & (t[:, 0] > 0)
d = d[ok]
print(f"{ok.sum()} waves; cycles per phase (s_memtime ticks), mean / median / p90")
for i, nm in enumerate(names):
    print(f"  {nm:22s} {d[:, i].mean():9.0f} {np.median(d[:, i]):9.0f} {np.percentile(d[:, i], 90):9.0f}")
tot = (t[ok, 6] - t[ok, 0])
print(f"  {'whole wave':22s} {tot.mean():9.0f} {np.median(tot):9.0f} {np.percentile(tot, 90):9.0f}")
print(f"  kernel span {t[ok, 6].max() - t[ok, 0].min()} ticks")
