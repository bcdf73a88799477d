"""Does the TV-L1 flow of the next frame pair overlap with a solve on another stream? (development aid)"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import api, ops, synth

K, rows, cols, ch, scale, iters = 9, 540, 960, 3, 4, 180
lr, flows, _ = synth.make_sequence(K, rows, cols, ch, scale, seed=1234, device="cuda")
y = [f.to(torch.float32) for f in lr]
gray = [f.to(torch.float32).mean(dim=2).contiguous() for f in lr]
solver, flower = api.BTV_L1(), api.BTV_L1()
flower.setOpticalFlow(api.DualTVL1())
solver.set("scale", scale); solver.set("iterations", iters)
flower.setOpticalFlow(api.DualTVL1())


def solve():
    ops.process_device(solver, y, flows, K // 2, wait=False)
    solver.synchronize()


def flow():
    flower.calcFlowDevice(gray[0], gray[1])


def timed(*fns):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ts = [threading.Thread(target=f) for f in fns]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def flows_side_by_side(n):
    def run():
        flower.calcFlowDevicePairs(gray[:n], gray[1:n + 1], in_flight=n)
    return run


solve(); flow()
for n in (1, 2, 3, 4):
    f = flows_side_by_side(n)
    f()
    alone, both = timed(f), timed(solve, f)
    print(f"{n} pairs side by side: alone {alone:.2f} ms ({alone / n:.2f} per pair), beside a solve {both:.2f} ms", flush=True)
print(f"solve alone {timed(solve):.2f} ms, flow alone {timed(flow):.2f} ms, both at once {timed(solve, flow):.2f} ms", flush=True)
