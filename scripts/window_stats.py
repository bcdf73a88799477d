"""How often does the forward kernel's LDS window hold every source? (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import api, ops, synth

rows, cols, ch, s, n = 540, 960, 3, 4, 9
F0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
MY, MX = int(sys.argv[1]) if len(sys.argv) > 1 else 15, int(sys.argv[2]) if len(sys.argv) > 2 else 16
lr, flows, _ = synth.make_sequence(F0 + n, rows, cols, ch, s, seed=1234, device="cuda")
lr, flows = lr[F0:], flows[F0:]
sr = api.BTV_L1()
sr.setOpticalFlow(api.DualTVL1())
sr.setOpticalFlow(api.DualTVL1())
gray = [f.to(torch.float32).mean(dim=2) for f in lr]
H, W = rows * s, cols * s


def stats(name, fl_list):
    rel = ops.relative_motions(fl_list, n // 2, rows, cols)
    for k in range(9):
        _, bwd = ops.build_maps(rel[k], s)
        bx, by = bwd[..., 0].long(), bwd[..., 1].long()
        dy = by - torch.arange(H, device=by.device).view(H, 1)
        dx = bx - torch.arange(W, device=bx.device).view(1, W)
        # per LR row: HR rows 4y-1..4y+2 (approximately rows 4y..4y+3 here), per strip of 252 HR columns
        ok = (dy.abs() <= MY) & (dx.abs() <= MX)
        nw = W // 252 * 252
        row_ok = ok[:, :nw].view(H // 4, 4, -1, 252).all(-1).all(1)          # (LR row, strip)
        g = by.view(H, W // 4, 4)
        differ = (g != g[..., :1]).any(-1)                                    # group has dy != 0
        grp_rows = differ[:, : nw // 4].view(H, -1, 63).any(-1)                # (HR row, strip)
        print(f"{name} frame {k}: |dy| max {dy.abs().max().item()} |dx| max {dx.abs().max().item()}  "
              f"rows*strips inside window {row_ok.float().mean().item()*100:.1f}%  "
              f"wave group-rows with differing source rows {grp_rows.float().mean().item()*100:.1f}%", flush=True)


stats("exact", flows)
est = []
for i in range(n - 1):
    f, it = sr.calcFlowDevice(gray[i], gray[i + 1])
    est.append(f)
stats("tvl1 ", est)
