"""Where does a streaming step spend its time when the flow comes from a Python plug-in? (development aid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import api, spans, synth, _lib as L

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--timing", type=int, default=0)
ap.add_argument("--span", type=int, default=0)
ap.add_argument("--calls", type=int, default=9)
ap.add_argument("--modes", default="device")
A = ap.parse_args()
rows, cols, ch, s, r, n = 540, 960, 3, 4, 4, 18
cams = synth.camera_path(n + 1, seed=1234)
frames = [synth.degrade(synth.ground_truth_frame(c, rows * s, cols * s, ch, 1234, "cuda"), s) for c in cams[:n]]
flows = [synth.true_forward_flow(cams[i], cams[i + 1], rows, cols, s, "cuda") for i in range(n - 1)]
torch.cuda.synchronize()
for mode in A.modes.split(","):
    fl = flows if mode == "device" else [f.cpu().numpy() for f in flows]
    sr = api.BTV_L1()
    sr.set("scale", s); sr.set("temporalAreaRadius", r); sr.set("outputMemory", L.SRX_MEM_DEVICE)
    st = {"i": 0, "t": 0.0}
    def cb(a, b):
        t0 = time.perf_counter()
        f = fl[st["i"]]; st["i"] += 1
        st["t"] += time.perf_counter() - t0
        return f
    sr.setOpticalFlow(cb)
    if A.span:
        spans.configure_span(sr, [], frames[:13], frames[13:17])
    else:
        sr.setFrameSource(spans.ListSource(frames))
    for k in range(A.calls):
        if k == 2:
            sr.set("kernelTiming", A.timing)
        st["t"] = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = sr.nextFrame()
        sr.synchronize()
        print(f"{mode} flows, call {k}: {1e3*(time.perf_counter()-t0):.1f} ms (callback body {1e3*st['t']:.2f} ms)", flush=True)
