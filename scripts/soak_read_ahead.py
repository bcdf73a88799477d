"""Full-size streaming with and without readAhead, both built-in flows: the frames must be identical (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import api, synth, _lib as L

n, rows, cols, T = int(sys.argv[1]) if len(sys.argv) > 1 else 24, 540, 960, 6
cams = synth.camera_path(n, seed=77)
frames = [synth.degrade(synth.ground_truth_frame(c, rows * 4, cols * 4, 3, 77, "cuda"), 4) for c in cams]


class Src:
    def __init__(self):
        self.i = 0

    def nextFrame(self):
        self.i += 1
        return frames[self.i - 1] if self.i <= n else None

    def reset(self):
        self.i = 0


def run(flow, ahead):
    sr = api.BTV_L1()
    sr.set("iterations", T)
    sr.set("readAhead", ahead)
    sr.set("outputMemory", L.SRX_MEM_DEVICE)
    sr.setOpticalFlow(flow)
    sr.setFrameSource(Src())
    out = []
    while True:
        f = sr.nextFrame()
        if f is None:
            break
        out.append(f.clone())
    return out


bad = 0
for name, flow in (("farneback", api.Farneback), ("tvl1", api.DualTVL1)):
    a, b = run(flow(), 0), run(flow(), 1)
    same = len(a) == len(b) == n and all(torch.equal(x, y) for x, y in zip(a, b))
    print(name, len(a), len(b), "identical" if same else "DIFFERENT", flush=True)
    if not same:
        print("  differing frames:", [(i, int((x != y).sum())) for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x, y)], flush=True)
    bad += not same
sys.exit(1 if bad else 0)
