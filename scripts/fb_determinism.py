"""Farneback flow of the same pair, repeated with other work in between, must not change (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import api, synth

rows, cols = 540, 960
cams = synth.camera_path(4, seed=77)
fr = [synth.degrade(synth.ground_truth_frame(c, rows * 4, cols * 4, 3, 77, "cuda"), 4) for c in cams]
sr = api.BTV_L1()
sr.setOpticalFlow(api.Farneback())
a = torch.from_numpy(sr.calcFlow(fr[0], fr[1]))
b = torch.from_numpy(sr.calcFlow(fr[2], fr[3]))
c = torch.from_numpy(sr.calcFlow(fr[0], fr[1]))
sr2 = api.BTV_L1()
sr2.setOpticalFlow(api.Farneback())
d = torch.from_numpy(sr2.calcFlow(fr[0], fr[1]))
print("repeat equal:", torch.equal(a, c), "fresh object equal:", torch.equal(a, d), "diff px:", int((a != c).any(-1).sum()), int((a != d).any(-1).sum()))
if not torch.equal(a, c):
    ys, xs = torch.nonzero((a != c).any(-1), as_tuple=True)
    print("rows", int(ys.min()), int(ys.max()), "cols", int(xs.min()), int(xs.max()), "max abs", float((a - c).abs().max()))
