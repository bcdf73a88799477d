"""Host time to enqueue one solve vs GPU time to run it (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import api, ops, synth

K, rows, cols, ch, scale = 9, 540, 960, 3, 4
lr, flows, _ = synth.make_sequence(K, rows, cols, ch, scale, seed=1234, device="cuda")
y = [f.to(torch.float32) for f in lr]
sr = api.BTV_L1()
sr.set("scale", scale)
for timing in (0, 1):
    sr.set("kernelTiming", timing)
    for iters in (60, 180):
        sr.set("iterations", iters)
        ops.process_device(sr, y, flows, K // 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.process_device(sr, y, flows, K // 2, wait=False)
        t1 = time.perf_counter()
        sr.synchronize()
        t2 = time.perf_counter()
        print(f"kernelTiming={timing} iterations={iters}: enqueue {1e3*(t1-t0):.2f} ms ({1e6*(t1-t0)/iters:.1f} us/iter), "
              f"total {1e3*(t2-t0):.2f} ms ({1e6*(t2-t0)/iters:.1f} us/iter)", flush=True)
