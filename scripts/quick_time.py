"""Quick wall-clock timing of one solve on synthetic device-resident inputs (development aid)."""
import argparse
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import api, ops, synth

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=540)
ap.add_argument("--cols", type=int, default=960)
ap.add_argument("--ch", type=int, default=3)
ap.add_argument("--scale", type=int, default=4)
ap.add_argument("--K", type=int, default=9)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--lam", type=float, default=0.03)
ap.add_argument("--shift", type=float, default=1.7, help="camera translation per frame, LR pixels")
ap.add_argument("--rot", type=float, default=0.002, help="camera rotation scale, radians")
ap.add_argument("--first", type=int, default=0, help="index of the window's first frame on the camera path")
a = ap.parse_args()

lr, flows, _ = synth.make_sequence(a.first + a.K, a.rows, a.cols, a.ch, a.scale, seed=1234, device="cuda",
                                   max_shift=a.shift, max_rot=a.rot)
lr, flows = lr[a.first:], flows[a.first:]
y = [f.to(torch.float32) for f in lr]
sr = api.BTV_L1()
sr.set("scale", a.scale)
sr.set("lambda", a.lam)
for iters in (1, a.iters):
    sr.set("iterations", iters)
    ts = []
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = ops.process_device(sr, y, flows, a.K // 2)
        ts.append(time.perf_counter() - t0)
    print(f"iterations={iters}: {min(ts)*1e3:.2f} ms (min of {a.reps})", flush=True)
    if iters == 1:
        t1 = min(ts)
per_iter = (min(ts) - t1) / (a.iters - 1)
P = a.rows * a.cols * a.scale ** 2
bytes_iter = 8 * a.ch * P + a.K * (8 * P + 4 * a.ch * P / a.scale ** 2)
print(f"per-iteration {per_iter*1e3:.3f} ms; algorithmic {bytes_iter/1e6:.1f} MB/iter -> {bytes_iter/per_iter/1e9:.1f} GB/s "
      f"({bytes_iter/per_iter/8e12*100:.1f}% of 8 TB/s); 180-iter frame est {per_iter*180*1e3:.1f} ms")
# per-kernel times from HIP events on the solver's stream
sr.set("kernelTiming", 1)
sr.set("iterations", a.iters)
ops.process_device(sr, y, flows, a.K // 2)
f, b, n = sr.kernelTimes()
print(f"kernels: forward {f/n*1e3:.1f} us, backproject {b/n*1e3:.1f} us per launch over {n} iterations "
      f"[{os.environ.get('SRX_LIBRARY', 'default library')}]", flush=True)
