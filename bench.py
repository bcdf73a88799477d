#!/usr/bin/env python3
"""Throughput of the BTV-L1 super-resolution hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one steady-state SuperResolution::nextFrame() on one GPU: one new
low-resolution frame enters the ring, its flow is attached, one output frame is
solved (180 BTV-L1 iterations over a 2*4+1 frame window) and returned as 8-bit.
With read-ahead (the default here, --no-read-ahead turns it off) the frame that
enters and its flow belong to the next call: they are fetched while this call's
solve runs; the work per step is the same.
Workload (the configuration the headline metric is quoted on): 960x540 RGB ->
3840x2160, scale 4, temporalAreaRadius 4.  Inputs are synthetic
(superresolution_amd/synth.py), 8-bit, resident in HBM before the timed region.

Multi-GPU: the stream is cut into contiguous spans of output frames, one per
rank (weak scaling: every rank solves `steps` frames); neighbouring ranks swap
the temporalAreaRadius boundary frames point to point before the timed region.
No collective sits on the data path.

Prints one JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=540, help="LR rows")
    ap.add_argument("--cols", type=int, default=960, help="LR cols")
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--scale", type=int, default=4)
    ap.add_argument("--radius", type=int, default=4)
    ap.add_argument("--iterations", type=int, default=180)
    ap.add_argument("--flow", choices=["synthetic", "tvl1", "farneback"], default="tvl1",
                    help="motion stage: exact synthetic motion resident in HBM, the built-in TV-L1 flow, or the "
                         "built-in Farneback flow (the reference's default flow class)")
    ap.add_argument("--no-read-ahead", action="store_true",
                    help="read each frame in the call that needs it, as the reference does (default: the next "
                         "frame is read one call early and its TV-L1 flow runs beside the current solve)")
    ap.add_argument("--host-io", action="store_true",
                    help="hand over host frames and take host output (the PCIe-inclusive rate; never the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not bracket the solver kernels with HIP events (the roofline object is then left out)")
    ap.add_argument("--cpu-iters", type=int, default=12, help="iterations of the CPU sample solve")
    return ap.parse_args()


def algorithmic_bytes(P, C, K, s):
    """Compulsory-traffic model per solver iteration: X read + write, one int16x2 index per HR pixel and
    map, one LR frame per window frame.  Split by kernel: forward = K * (4P + 4CP/s^2), back-projection +
    update = 8CP + K * 4P."""
    fwd = K * (4 * P + 4 * C * P / s ** 2)
    bwd = 8 * C * P + K * 4 * P
    return fwd, bwd


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes (profiles/*_traffic.json, written by
    scripts/pmc_traffic.py on the same workload), or None."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", "*_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        for name, v in d["kernels"].items():
            if name.startswith(kernel):
                return v["traffic_bytes_per_launch"], os.path.relpath(files[-1], here)
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def cpu_baseline(args, lr_u8, flows, base):
    """The unfused restatement of the reference CPU path (oracle/, OpenMP over rows) on one window of the same
    workload, for a bounded number of iterations, scaled to the full iteration count; at all host threads the
    process may use and, on a shorter sample, at one thread."""
    import oracle

    y = [f.cpu().numpy().astype(np.float32) for f in lr_u8]
    fl = [f.cpu().numpy() for f in flows]
    mpix = args.rows * args.cols * args.scale ** 2 / 1e6

    def measure(n_iters):
        times = {}
        for it in (1, 1 + n_iters):
            p = oracle.BtvParams(scale=args.scale, iterations=it)
            t0 = time.perf_counter()
            oracle.btvl1_process(y, fl, base, p)
            times[it] = time.perf_counter() - t0
        per_iter = (times[1 + n_iters] - times[1]) / n_iters
        setup = max(times[1] - per_iter, 0.0)
        return per_iter, setup, mpix / (setup + per_iter * args.iterations)

    threads = oracle.max_threads()
    per_iter, setup, value = measure(args.cpu_iters)
    out = {
        "value": value,
        "unit": "Mpix/s",
        "cores": threads,
        "kind": "port",
        "sample": f"one {len(y)}-frame window of the same workload, {1 + args.cpu_iters} solver iterations timed "
                  f"({per_iter:.2f} s/iteration, {setup:.2f} s set-up), scaled to {args.iterations} iterations",
    }
    if threads > 1 and args.cpu_iters >= 4:
        oracle.set_threads(1)
        try:
            p1, s1, v1 = measure(1)
        finally:
            oracle.set_threads(threads)
        out["single_thread"] = {"value": v1, "unit": "Mpix/s", "cores": 1,
                                "sample": f"same window, 2 solver iterations timed ({p1:.2f} s/iteration)"}
    return out


def copy_ceiling_gbs(dev) -> float:
    """Read + write rate of a plain device-to-device copy of 1 GiB (outside the timed region): what a kernel
    that only streams can reach on this GPU, to set beside the 8 TB/s spec peak."""
    n = 1 << 28
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * 4 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SRX_BENCH_REHEARSE=1: all ranks share GPU 0 and talk over gloo with host copies of the halo - a way to
    # walk the multi-rank code on a one-GPU box; its numbers mean nothing.
    rehearse = os.environ.get("SRX_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from superresolution_amd import api, spans, synth, _lib as L

    r, s = args.radius, args.scale
    # Calls: one set-up call (ring warm-up: 2r+2 reads), then W warm-up and K timed steady-state steps with one
    # read, one flow and one solve each.  The solve of a call runs r+1 frames ahead of the output it returns, so
    # a span of W + K + r + 2 owned frames keeps every timed solve inside the span with a full window.
    read_ahead = args.flow != "synthetic" and not args.no_read_ahead
    n_own = args.warmup + args.steps + r + 2 + (1 if read_ahead else 0)  # read-ahead pulls one frame more
    total = n_own * world
    a = rank * n_own
    cams = synth.camera_path(total + 2 * r + 2, seed=1234)

    def frame(t):
        hr = synth.ground_truth_frame(cams[t], args.rows * s, args.cols * s, args.channels, 1234, dev)
        return synth.degrade(hr, s)

    own = [frame(t) for t in range(a, a + n_own)]
    if world > 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if rehearse:
            left, right = spans.exchange_halo([f.cpu() for f in own], r, rank, world)
            left, right = [f.to(dev) for f in left], [f.to(dev) for f in right]
        else:
            left, right = spans.exchange_halo(own, r, rank, world)
        torch.cuda.synchronize()
        halo_ms = (time.perf_counter() - t0) * 1e3
    else:
        left, right, halo_ms = [], [], 0.0
    if rank == world - 1:
        # the stream continues past the benchmarked frames, so that every timed step sees a full window
        right = [frame(t) for t in range(a + n_own, a + n_own + r)]
    first = a - len(left)
    frames = left + own + right
    flows = [synth.true_forward_flow(cams[first + i], cams[first + i + 1], args.rows, args.cols, s, dev)
             for i in range(len(frames) - 1)]

    sr = api.BTV_L1(local_rank)
    sr.set("scale", s)
    sr.set("iterations", args.iterations)
    sr.set("temporalAreaRadius", r)
    sr.set("outputMemory", L.SRX_MEM_HOST if args.host_io else L.SRX_MEM_DEVICE)
    sr.set("readAhead", 1 if read_ahead else 0)
    if args.flow == "tvl1":
        sr.setOpticalFlow(api.DualTVL1())
    elif args.flow == "farneback":
        sr.setOpticalFlow(api.Farneback())
    else:
        counter = {"i": 0}

        def flow_cb(_f0, _f1):
            f = flows[counter["i"]]
            counter["i"] += 1
            return f

        sr.setOpticalFlow(flow_cb)
    if args.host_io:
        left, own, right = ([f.cpu().numpy() for f in v] for v in (left, own, right))
    spans.configure_span(sr, left, own, right)

    def step():
        out = sr.nextFrame()
        if out is None:
            raise RuntimeError("stream ended inside the benchmark")
        return out

    step()  # ring warm-up: 2r+2 reads and the first solves (not a steady-state step)
    for _ in range(args.warmup):
        step()
    sr.set("kernelTiming", 0 if args.no_kernel_timing else 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        out = step()
        step_ms.append((time.perf_counter() - ts) * 1e3)
    sr.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    fwd_ms, bwd_ms, n_it = sr.kernelTimes()
    sr.set("kernelTiming", 0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        P = args.rows * s * args.cols * s
        K = 2 * r + 1
        mpix = P / 1e6
        bytes_fwd, bytes_bwd = algorithmic_bytes(P, args.channels, K, s)
        bwd_launch_s = max(bwd_ms / max(n_it, 1) / 1e3, 1e-12)
        fwd_launch_s = max(fwd_ms / max(n_it, 1) / 1e3, 1e-12)
        default_workload = (args.rows, args.cols, args.channels, s, r) == (540, 960, 3, 4, 4)
        kernels = {}
        # which kernels the library picks for this scale (default parameters otherwise)
        fwd_name = {4: "k_forward_sign_lds", 2: "k_forward_sign_s2"}.get(s, "k_forward_sign_tile")
        bwd_name = {4: "k_backproject_update_s4", 2: "k_backproject_update_s4"}.get(s, "k_backproject_update")
        for name, what, nbytes, secs in (
                (fwd_name, "sign(y_k - D H M_k X) of all window frames", bytes_fwd, fwd_launch_s),
                (bwd_name, "back-projection of all window frames + BTV + update", bytes_bwd, bwd_launch_s)):
            traffic, traffic_src = measured_traffic(name) if default_workload else (None, None)
            kernels[name] = {"kernel": f"{name} ({what})", "achieved": nbytes / secs / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": nbytes / secs / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                             "traffic_source": traffic_src, "algorithmic_bytes_per_launch": nbytes,
                             "avg_launch_us": secs * 1e6}
        # the dominant kernel is the one the solver spends more time in
        dominant = fwd_name if fwd_launch_s >= bwd_launch_s else bwd_name
        other = bwd_name if dominant == fwd_name else fwd_name
        line = {
            "metric": "output Mpix/s, BTV-L1 4x 960x540->4K, 1/2/4/8 GPUs; % HBM roofline",
            "value": world * args.steps * mpix / elapsed,
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{'configs[3]' if default_workload else 'other configuration'}: {args.cols}x{args.rows} {'RGB' if args.channels == 3 else 'gray'} -> "
                            f"{args.cols * s}x{args.rows * s}, {s}x BTV-L1, {args.iterations} iterations, "
                            f"temporalAreaRadius {r} (window {K}), streaming nextFrame(), motion: "
                            + {"tvl1": "built-in TV-L1 flow", "farneback": "built-in Farneback flow"}.get(
                                args.flow, "exact synthetic flow resident in HBM"),
                "frames_per_gpu": args.steps,
                "io": "host frames in, host frames out (PCIe inclusive)" if args.host_io else "frames resident in HBM",
                "read_ahead": read_ahead,
                "partition": "contiguous span of output frames per GPU, halo of r LR frames swapped point to point",
                "halo_exchange_ms": round(halo_ms, 3),
                "step_ms": [round(v, 2) for v in step_ms],
                "delivered_frame": f"{args.cols * s - 14}x{args.rows * s - 14} (cropped by btvKernelSize)",
            },
            "roofline": dict(
                {"bound": "hbm"}, **kernels[dominant],
                other_kernel=kernels[other],
                iteration={"algorithmic_bytes": bytes_fwd + bytes_bwd,
                           "achieved": (bytes_fwd + bytes_bwd) / (fwd_launch_s + bwd_launch_s) / 1e9,
                           "frac": (bytes_fwd + bytes_bwd) / (fwd_launch_s + bwd_launch_s) / 1e9 / HBM_PEAK_GBS}),
        }
        if args.no_kernel_timing:
            del line["roofline"]
        else:
            ceiling = copy_ceiling_gbs(dev)
            line["roofline"]["copy_ceiling"] = {"value": round(ceiling, 1), "unit": "GB/s",
                                                "what": "device-to-device copy of 1 GiB, read + write bytes",
                                                "frac_of_ceiling": line["roofline"]["achieved"] / ceiling}
        if world == 1 and not args.no_cpu_baseline:
            base = min(r, len(frames) - 1)
            win = frames[:2 * r + 1]
            line["cpu_baseline"] = cpu_baseline(args, win, flows[:2 * r], base)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
