"""Row kinds and per-block times of the LAST forward launch of a streamed clip (built-in TV-L1 flow, the benchmark's
clip) from a -DSRX_STAMPS build (development aid).
usage: SRX_LIBRARY=variants/libsrx_stamps.so python scripts/stamp_stream.py [frames]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from superresolution_amd import api, spans, synth, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rows, cols, ch, s = 540, 960, 3, 4
dev = torch.device("cuda", 0)
cams = synth.camera_path(n + 12, seed=1234)
frames = [synth.degrade(synth.ground_truth_frame(cams[t], rows * s, cols * s, ch, 1234, dev), s) for t in range(n)]
sr = api.BTV_L1()
sr.set("iterations", 2)
sr.set("lambda", 0.0)
sr.set("outputMemory", _lib.SRX_MEM_DEVICE)
sr.setOpticalFlow(api.DualTVL1() if "--farneback" not in sys.argv else api.Farneback())
sr.setFrameSource(spans.ListSource(frames))
outs = 0
while outs < n - 6 and sr.nextFrame() is not None:
    outs += 1
sr.synchronize()
lib = _lib.lib()
n_waves = 256 * 9
buf = np.zeros(n_waves * 8, dtype=np.uint64)
assert lib.srx_debug_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size)) == 0
bw = buf.reshape(256, 9, 8).astype(np.int64)
whole = bw[:, :, 4].max(axis=1)
kinds = np.stack([(bw[:, :, 7] >> (16 * i)) & 0xFFFF for i in range(4)], axis=-1)
print("outputs", outs, "block ticks: mean", whole.mean(), "p50", np.median(whole), "max", whole.max())
print("rows by kind (mixed, inside, outside, identity) by wave:")
for w in range(9):
    print("  ", w, kinds[:, w, :].sum(axis=0))
order = np.argsort(-whole)[:10]
for b in order:
    tile = (b & 7) * 32 + (b >> 3)
    print("  block", b, "tile (strip, row group)", (int(tile % 16), int(tile // 16)), "ticks", whole[b], "mixed rows by wave", kinds[b, :, 0],
          "outside", kinds[b, :, 2])
