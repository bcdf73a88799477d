"""Statistics of the backward warp map built from TV-L1 flows vs exact flows (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import api, ops, synth

rows, cols, ch, s, n = 540, 960, 3, 4, 5
lr, flows, _ = synth.make_sequence(n, rows, cols, ch, s, seed=1234, device="cuda")
sr = api.BTV_L1()
sr.setOpticalFlow(api.DualTVL1())
sr.setOpticalFlow(api.DualTVL1())
gray = [f.to(torch.float32).mean(dim=2) for f in lr]


def stats(name, fl_list):
    rel = ops.relative_motions(fl_list, 0, rows, cols)   # motions relative to frame 0
    for k in (1, len(rel) - 1):
        fwd, bwd = ops.build_maps(rel[k], s)
        torch.cuda.synchronize()
        bx, by = bwd[..., 0].long(), bwd[..., 1].long()
        H, W = rows * s, cols * s
        inside = (bx >= 0) & (bx < W) & (by >= 0) & (by < H)
        idx = torch.where(inside, by * W + bx, torch.full_like(bx, H * W))
        g = idx.view(H, W // 4, 4)
        d = g - g[..., :1] - torch.arange(4, device=g.device)
        cons = (d == 0).all(dim=-1)
        dxs = (bx.view(H, W // 4, 4) - bx.view(H, W // 4, 4)[..., :1] - torch.arange(4, device=g.device))
        dys = (by.view(H, W // 4, 4) - by.view(H, W // 4, 4)[..., :1])
        xj = (~cons) & (dys == 0).all(-1) & (dxs.abs() <= 1).all(-1)
        small = (~cons) & (dys.abs() <= 15).all(-1) & (dxs.abs() <= 15).all(-1)
        # two-run: every pixel continues the run from the first or ends the run at the last
        dl = d
        dr = g - g[..., 3:] + (3 - torch.arange(4, device=g.device))
        two = (~cons) & ((dl == 0) | (dr == 0)).all(-1)
        wave_rows = cons.view(H, -1)  # per row of groups
        # a wave row = 64 consecutive groups (63 owners): fraction of wave rows with any broken group
        nw = (W // 4) // 64 * 64
        any_broken = (~cons[:, :nw]).view(H, -1, 64).any(-1).float().mean().item()
        # LR pixels whose 5x5 HR neighbourhood is warped by one integer translation
        idx2 = idx.view(H, W)
        off = idx2 - (torch.arange(H, device=idx.device).view(H, 1) * W + torch.arange(W, device=idx.device).view(1, W))
        off = torch.where(inside, off, torch.full_like(off, -(1 << 40)))
        import torch.nn.functional as F
        o4 = off.float().view(1, 1, H, W)
        mx = F.max_pool2d(o4, 5, stride=4, padding=2)      # window centred on (4y, 4x)
        mn = -F.max_pool2d(-o4, 5, stride=4, padding=2)
        centre = off[::4, ::4].float()
        clean = ((mx[0, 0] == mn[0, 0]) & (centre > -1e12))
        clean_rows = clean.float().mean().item()
        wave_clean = clean[:, :(W // 4) // 63 * 63].reshape(H // 4, -1, 63).all(-1).float().mean().item()
        print(f"   pure-translation 5x5 patches: {clean_rows*100:.2f}% of LR pixels; 63-px wave rows entirely clean: {wave_clean*100:.1f}%")
        print(f"{name} frame {k}: consecutive groups {cons.float().mean().item()*100:.2f}%  broken: x-jitter(|dx|<=1,dy=0) "
              f"{xj.float().mean().item()*100:.2f}%  two-run {two.float().mean().item()*100:.2f}%  |d|<=15 "
              f"{small.float().mean().item()*100:.2f}%  wave-rows with a broken group {any_broken*100:.1f}%", flush=True)


stats("exact flow ", flows)
est = []
for i in range(n - 1):
    f, it = sr.calcFlowDevice(gray[i], gray[i + 1])
    est.append(f)
stats("TV-L1 flow ", est)
