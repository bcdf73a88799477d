"""Times the built-in Farneback flow on device-resident grey pairs (no oracle involved)."""
import sys, time
import numpy as np
import torch
from superresolution_amd import api, synth

for rows, cols in ((540, 960), (1080, 1920)):
    lr, _, _ = synth.make_sequence(2, rows, cols, 1, 4, seed=3)
    a, b = (f.float().cuda().reshape(rows, cols) for f in lr)
    sr = api.BTV_L1()
    sr.setOpticalFlow(api.Farneback())
    sr.calcFlowDevice(a, b)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 5
    for _ in range(n):
        sr.calcFlowDevice(a, b)
    torch.cuda.synchronize()
    print(f"{rows}x{cols}: {(time.perf_counter() - t) / n * 1e3:.2f} ms per pair", flush=True)
