#!/usr/bin/env python3
"""TV-L1 pairs side by side (srx_flow_calc_device_pairs): the flows must be those of pair-after-pair calls; time per
pair for 1 .. 4 pairs in flight.  usage: flow_pairs_time.py [rows cols [pairs]]"""
import sys
import time

import torch

from superresolution_amd import api, synth

rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dev = torch.device("cuda:0")
cams = synth.camera_path(n + 1, seed=4321)
gray = [synth.ground_truth_frame(cams[t], rows, cols, 1, 4321, dev)[:, :, 0].to(torch.float32).contiguous()
        for t in range(n + 1)]
sr = api.BTV_L1(0)
sr.setOpticalFlow(api.DualTVL1())
single = [sr.calcFlowDevice(gray[t], gray[t + 1]) for t in range(n)]
for k in (1, 2, 3, 4):
    for rep in range(3):
        t0 = time.perf_counter()
        flows, its = sr.calcFlowDevicePairs(gray[:-1], gray[1:], in_flight=k)
        dt = time.perf_counter() - t0
    same = all(torch.equal(f, s[0]) and i == s[1] for f, i, s in zip(flows, its, single))
    print(f"in flight {k}: {dt / n * 1e3:.3f} ms per pair, {n / dt:.1f} pairs/s, identical to single calls: {same}", flush=True)
    assert same
