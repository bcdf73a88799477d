"""Do two independent solves on two handles (two HIP streams) overlap? (development aid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import api, ops, synth

K, rows, cols, ch, scale, iters = 9, 540, 960, 3, 4, 40
lr, flows, _ = synth.make_sequence(K, rows, cols, ch, scale, seed=1234, device="cuda")
y = [f.to(torch.float32) for f in lr]
hs = [api.BTV_L1() for _ in range(3)]
for h in hs:
    h.set("scale", scale); h.set("iterations", iters)
    ops.process_device(h, y, flows, K // 2)
    h.synchronize()
torch.cuda.synchronize()
for n in (1, 2, 3):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [ops.process_device(h, y, flows, K // 2, wait=False) for h in hs[:n]]
        for h in hs[:n]:
            h.synchronize()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{n} concurrent solves of {iters} iterations: {best*1e3:.2f} ms -> {best/n/iters*1e6:.1f} us per iteration per solve", flush=True)
