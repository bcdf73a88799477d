"""Full-size streaming: host frames vs device frames, host output vs device output, reset() in mid-stream, two
handles interleaved - all must give the same frames (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import api, synth, _lib as L

n, rows, cols, T = 14, 540, 960, 5
cams = synth.camera_path(n, seed=78)
dev = [synth.degrade(synth.ground_truth_frame(c, rows * 4, cols * 4, 3, 78, "cuda"), 4) for c in cams]
host = [f.cpu().numpy() for f in dev]


class Src:
    def __init__(self, frames):
        self.f, self.i = frames, 0

    def nextFrame(self):
        self.i += 1
        return self.f[self.i - 1] if self.i <= len(self.f) else None

    def reset(self):
        self.i = 0


def make(frames, ahead, out_dev):
    sr = api.BTV_L1()
    sr.set("iterations", T)
    sr.set("readAhead", ahead)
    sr.set("outputMemory", L.SRX_MEM_DEVICE if out_dev else L.SRX_MEM_HOST)
    sr.setOpticalFlow(api.Farneback())
    sr.setFrameSource(Src(frames))
    return sr


def take(f):
    return f.cpu().numpy().copy() if hasattr(f, "cpu") else np.array(f, copy=True)


def drain(sr, limit=None):
    out = []
    while limit is None or len(out) < limit:
        f = sr.nextFrame()
        if f is None:
            break
        out.append(take(f))
    return out


ref = drain(make(dev, 0, True))
ok = len(ref) == n
for name, frames, ahead, od in (("host in, host out", host, 0, False), ("host in, ahead", host, 1, False),
                                ("device in, host out, ahead", dev, 1, False), ("host in, device out, ahead", host, 1, True)):
    got = drain(make(frames, ahead, od))
    same = len(got) == n and all(np.array_equal(a, b) for a, b in zip(ref, got))
    print(name, "identical" if same else "DIFFERENT", flush=True)
    ok &= same
sr = make(dev, 1, True)
part = drain(sr, 5)
sr.reset()
again = drain(sr)
same = all(np.array_equal(a, b) for a, b in zip(ref, part)) and len(again) == n and all(np.array_equal(a, b) for a, b in zip(ref, again))
print("reset in mid-stream", "identical" if same else "DIFFERENT", flush=True)
ok &= same
a, b = make(dev, 1, True), make(host, 0, False)
oa, ob = [], []
for _ in range(n):
    oa.append(take(a.nextFrame()))
    ob.append(take(b.nextFrame()))
same = all(np.array_equal(x, y) for x, y in zip(ref, oa)) and all(np.array_equal(x, y) for x, y in zip(ref, ob))
print("two handles interleaved", "identical" if same else "DIFFERENT", flush=True)
ok &= same
sys.exit(0 if ok else 1)
