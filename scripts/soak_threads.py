"""Two handles driven from two threads at once on one GPU must give the frames of a single-threaded run
(development aid)."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import api, synth, _lib as L

n, rows, cols, T = 12, 270, 480, 8
cams = synth.camera_path(n, seed=79)
dev = [synth.degrade(synth.ground_truth_frame(c, rows * 4, cols * 4, 3, 79, "cuda"), 4) for c in cams]
torch.cuda.synchronize()


class Src:
    def __init__(self):
        self.i = 0

    def nextFrame(self):
        self.i += 1
        return dev[self.i - 1] if self.i <= n else None

    def reset(self):
        self.i = 0


def run(flow, ahead, out):
    sr = api.BTV_L1()
    sr.set("iterations", T)
    sr.set("readAhead", ahead)
    sr.setOpticalFlow(flow())
    sr.setFrameSource(Src())
    while True:
        f = sr.nextFrame()
        if f is None:
            break
        out.append(np.array(f, copy=True))


ref_f, ref_t = [], []
run(api.Farneback, 0, ref_f)
run(api.DualTVL1, 0, ref_t)
ok = True
for rep in range(3):
    a, b = [], []
    ta = threading.Thread(target=run, args=(api.Farneback, 1, a))
    tb = threading.Thread(target=run, args=(api.DualTVL1, rep % 2, b))
    ta.start(), tb.start()
    ta.join(), tb.join()
    same = len(a) == n and len(b) == n and all(np.array_equal(x, y) for x, y in zip(ref_f, a)) and \
        all(np.array_equal(x, y) for x, y in zip(ref_t, b))
    print("round", rep, "identical" if same else "DIFFERENT", flush=True)
    ok &= same
sys.exit(0 if ok else 1)
