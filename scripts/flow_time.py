"""Wall-clock of the built-in TV-L1 flow on synthetic device-resident grey pairs (development aid)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1080)
ap.add_argument("--cols", type=int, default=1920)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--eps", type=float, default=0.01)
a = ap.parse_args()
cams = synth.camera_path(2, seed=5)
f = [synth.ground_truth_frame(c, a.rows, a.cols, 1, 5, "cuda")[:, :, 0].float().contiguous() for c in cams]
sr = api.BTV_L1()
sr.setOpticalFlow(api.DualTVL1())
sr.flowSet("epsilon", a.eps)
for _ in range(a.reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    flow, n = sr.calcFlowDevice(f[0], f[1])
    dt = time.perf_counter() - t0
    px = a.rows * a.cols
    print(f"{a.cols}x{a.rows}: {dt*1e3:.1f} ms, {n} inner iterations, mean flow {flow.mean(dim=(0,1)).tolist()}", flush=True)
