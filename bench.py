#!/usr/bin/env python3
"""Throughput of the BTV-L1 super-resolution hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg4|cfg2|cfg1|cfg5|tvl1]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no WORLD_SIZE in the environment, this process starts the N ranks itself
(`python -m torch.distributed.run`, child processes, before anything here touches a GPU) and exits with
their status.  A rank whose WORLD_SIZE differs from --gpus refuses to run.

One step = one steady-state SuperResolution::nextFrame() on one GPU: one new low-resolution frame enters
the ring, its flow is attached, one output frame is solved (180 BTV-L1 iterations over a 2*4+1 frame
window) and returned as 8-bit.  With read-ahead (the default here, --no-read-ahead turns it off) the frame
that enters and its flow belong to the next call: they are fetched while this call's solve runs; the work
per step is the same.
Default workload (the configuration the headline metric is quoted on, configs[3]): 960x540 RGB ->
3840x2160, scale 4, temporalAreaRadius 4.  Inputs are synthetic (superresolution_amd/synth.py), 8-bit,
resident in HBM before the timed region.

Multi-GPU: the stream is cut into contiguous spans of output frames, one per rank (weak scaling: every
rank solves `steps` frames); neighbouring ranks swap the temporalAreaRadius boundary frames point to
point before the timed region.  No collective sits on the data path.  `value` is the steady-state rate;
`config.span_inclusive` is the rate with every rank's first call (the ring warm-up: r+2 solves on the rank
that holds the start of the stream, fewer on the others) and the warm-up steps inside the clock.

--scaling strong --frames N (any --gpus): ONE clip of N output frames is cut into contiguous spans by
srx_span_plan; every rank streams its span from its first call (ring warm-up, the stream's first / last windows
on the ranks that hold them) to its last output.  `value` = N x Mpix / wall time (max over ranks), so the
imbalance of the spans and every rank's warm-up are inside it; with --with-single rank 0 first streams the whole
clip alone and the line carries `efficiency` = T1 / (n Tn).  The boundary frames are posted point to point as
soon as they exist, the rest of the span is synthesised while they travel, and the wait in front of the first
call is reported as the exposed halo time.

--workload tvl1 measures the flow stage alone (configs[2]): 1920x1080 grey frame pairs, 5 scales x 5 warps
x 300 inner iterations at most, pairs sharded over the ranks (independent, no exchange).

Prints one JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
METRIC = "output Mpix/s, BTV-L1 4x 960x540->4K, 1/2/4/8 GPUs; % HBM roofline"

# rows, cols, channels, scale, radius, which entry of BASELINE.json's configs
WORKLOADS = {
    "cfg1": (240, 320, 1, 4, 4, "configs[0]"),
    "cfg2": (270, 480, 3, 4, 4, "configs[1]"),
    "cfg4": (540, 960, 3, 4, 4, "configs[3]"),
    "cfg5": (1080, 1920, 3, 2, 8, "configs[4]"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["tvl1"], default="cfg4")
    ap.add_argument("--rows", type=int, default=None, help="LR rows (overrides the workload's)")
    ap.add_argument("--cols", type=int, default=None, help="LR cols")
    ap.add_argument("--channels", type=int, default=None)
    ap.add_argument("--scale", type=int, default=None)
    ap.add_argument("--radius", type=int, default=None)
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
                    help="do not time the solver kernels with HIP events (the roofline object is then left out)")
    ap.add_argument("--cpu-iters", type=int, default=12, help="iterations of the CPU sample solve")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when this process starts the ranks")
    ap.add_argument("--growing-rotation", action="store_true",
                    help="stress clip: the camera's rotation amplitude grows with the square root of the frame index "
                         "(corner displacements of tens of HR pixels between neighbouring frames late in the clip)")
    ap.add_argument("--pairs-in-flight", type=int, default=3,
                    help="tvl1 workload: independent frame pairs the library works on side by side (1 .. 4)")
    ap.add_argument("--pairs-per-step", type=int, default=0,
                    help="tvl1 workload: pairs of one step = one library call (default: twice --pairs-in-flight)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank solves --steps frames (the default); strong: one clip of --frames outputs "
                         "is cut into one span per rank")
    ap.add_argument("--frames", type=int, default=0, help="outputs of the whole clip (strong scaling); default: "
                                                         "(steps + warmup) x gpus, at least what the span plan needs")
    ap.add_argument("--with-single", action="store_true",
                    help="strong scaling: rank 0 first streams the whole clip alone; the line then carries efficiency")
    ap.add_argument("--single-ms", type=float, default=0.0,
                    help="strong scaling: wall time of the same clip on one GPU from an earlier run (for efficiency)")
    a = ap.parse_args()
    if a.workload != "tvl1":
        w = WORKLOADS[a.workload]
        a.default_shape = all(v is None for v in (a.rows, a.cols, a.channels, a.scale, a.radius))
        a.rows = w[0] if a.rows is None else a.rows
        a.cols = w[1] if a.cols is None else a.cols
        a.channels = w[2] if a.channels is None else a.channels
        a.scale = w[3] if a.scale is None else a.scale
        a.radius = w[4] if a.radius is None else a.radius
        a.config_name = w[5] if a.default_shape else "other configuration"
    else:
        a.rows = 1080 if a.rows is None else a.rows
        a.cols = 1920 if a.cols is None else a.cols
    return a


def launch_ranks(args) -> int:
    """Start args.gpus ranks of this script as child processes (one per GPU, RCCL) and return their status.
    Nothing in this process has touched a GPU yet; the children are fresh interpreters."""
    import socket

    port = args.master_port
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def algorithmic_bytes(P, C, K, s):
    """Compulsory-traffic model per solver iteration: X read + write, one int16x2 index per HR pixel and
    map, one LR frame per window frame.  Split by kernel: forward = K * (4P + 4CP/s^2), back-projection +
    update = 8CP + K * 4P."""
    fwd = K * (4 * P + 4 * C * P / s ** 2)
    bwd = 8 * C * P + K * 4 * P
    return fwd, bwd


def measured_traffic(kernel, tag=""):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes (profiles/rNN_traffic.json for the
    headline workload, profiles/rNN_<tag>_traffic.json for the others; written by scripts/pmc_traffic.py on the
    same workload), or None."""
    import glob
    import re
    pat = re.compile(r"r\d+_" + (tag + "_" if tag else "") + r"traffic\.json$")
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")) if pat.search(os.path.basename(f)))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        for name, v in d["kernels"].items():
            if name.startswith(kernel):
                return v["traffic_bytes_per_launch"], os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def cpu_baseline(args, lr_u8, flows, base):
    """The unfused restatement of the reference CPU path (oracle/, OpenMP over rows) on one window of the same
    workload, for a bounded number of iterations, scaled to the full iteration count; at all host threads the
    process may use and, on a shorter sample, at one thread."""
    import numpy as np
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
    # the sample doubles as a parity check at the benchmark's own size: the same window, the same number of
    # iterations, through the library's default kernels - the two results must be the same bits
    checked = None
    if lr_u8[0].is_cuda:
        import torch
        from superresolution_amd import api, ops

        p = oracle.BtvParams(scale=args.scale, iterations=1 + args.cpu_iters)
        want = oracle.btvl1_process(y, fl, base, p)
        sr = api.BTV_L1(lr_u8[0].device.index)
        sr.set("scale", args.scale)
        sr.set("iterations", 1 + args.cpu_iters)
        got = ops.process_device(sr, [f.to(torch.float32) for f in lr_u8], list(flows), base).cpu().numpy()
        checked = bool(np.array_equal(want.reshape(got.shape), got))
        if not checked:
            raise RuntimeError("cpu_baseline: the oracle's window differs from the GPU's for the same inputs")
    out = {
        "value": value,
        "unit": "Mpix/s",
        "cores": threads,
        "kind": "port",
        "sample": f"one {len(y)}-frame window of the same workload, {1 + args.cpu_iters} solver iterations timed "
                  f"({per_iter:.2f} s/iteration, {setup:.2f} s set-up), scaled to {args.iterations} iterations",
        "equals_gpu_output": checked,
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


def copy_ceilings(dev):
    """Read + write rate of plain device-to-device copies of 1 GiB (outside the timed region): the library's own
    16-bytes-per-lane copy kernel (srx_k_copy) and torch's copy_.  What a kernel that only streams reaches on
    this GPU, to set beside the 8 TB/s spec peak."""
    import torch
    from superresolution_amd import _lib as L

    n = 1 << 28
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    lib = L.lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    out = {}
    for name, fn in (("own_float4_kernel", lambda: L.check(lib.srx_k_copy(src.data_ptr(), dst.data_ptr(), 4 * n, st))),
                     ("torch_copy", lambda: dst.copy_(src))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        out[name] = 2.0 * 4 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    return out


class Ranks:
    """World bookkeeping of one rank: RCCL on one GPU per rank; SRX_BENCH_REHEARSE=1: all ranks share GPU 0 and
    talk over gloo with host copies of the halo; SRX_BENCH_REHEARSE=dry: no GPU at all, dry-run handles (host
    bookkeeping only) - ways to walk the multi-rank code where there is one GPU or none; their numbers mean
    nothing."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist

        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        mode = os.environ.get("SRX_BENCH_REHEARSE", "")
        self.rehearse = mode in ("1", "dry")
        self.dry = mode == "dry"
        if self.world != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {self.world}: refusing to report a line whose "
                  f"n_gpus differs from what was asked for", file=sys.stderr)
            sys.exit(2)
        if self.rehearse:
            self.local_rank = 0
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.rehearse:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
        if self.dry:
            self.dev = torch.device("cpu")
        else:
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
        self.dist, self.torch = dist, torch

    def sync(self):
        if not self.dry:
            self.torch.cuda.synchronize(self.dev)
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, seconds: float) -> float:
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device="cpu" if self.rehearse else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def finish(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def bench_solver(args):
    R = Ranks(args)
    torch = R.torch
    from superresolution_amd import api, spans, synth, _lib as L

    r, s = args.radius, args.scale
    world, rank, dev = R.world, R.rank, R.dev
    # Calls: one set-up call (ring warm-up: 2r+2 reads), then W warm-up and K timed steady-state steps with one
    # read, one flow and one solve each.  The solve of a call runs r+1 frames ahead of the output it returns, so
    # a span of W + K + r + 2 owned frames keeps every timed solve inside the span with a full window.
    read_ahead = args.flow != "synthetic" and not args.no_read_ahead and not R.dry
    n_own = args.warmup + args.steps + r + 2 + (1 if read_ahead else 0)  # read-ahead pulls one frame more
    total = n_own * world
    a, b, in_first, in_last = spans.span_plan(total, r, world, rank)
    assert b - a + 1 == n_own
    cams = synth.camera_path(total + 2 * r + 2, seed=1234, rot_growth=args.growing_rotation)

    def frame(t):
        hr = synth.ground_truth_frame(cams[t], args.rows * s, args.cols * s, args.channels, 1234, dev)
        return synth.degrade(hr, s)

    own = [frame(t) for t in range(a, a + n_own)]
    halo_ms = 0.0
    left, right = [], []
    if world > 1:
        R.sync()
        t0 = time.perf_counter()
        if R.rehearse and not R.dry:
            left, right = spans.exchange_halo([f.cpu() for f in own], r, rank, world)
            left, right = [f.to(dev) for f in left], [f.to(dev) for f in right]
        else:
            left, right = spans.exchange_halo(own, r, rank, world)
        if not R.dry:
            torch.cuda.synchronize(dev)
        halo_ms = (time.perf_counter() - t0) * 1e3
        assert len(left) == a - in_first and len(right) == in_last - b
    if rank == world - 1:
        # the stream continues past the benchmarked frames, so that every timed step sees a full window
        right = [frame(t) for t in range(a + n_own, a + n_own + r)]
    first = a - len(left)
    frames = left + own + right
    flows = [synth.true_forward_flow(cams[first + i], cams[first + i + 1], args.rows, args.cols, s, dev)
             for i in range(len(frames) - 1)]

    sr = api.BTV_L1(-1 if R.dry else R.local_rank)
    sr.set("scale", s)
    sr.set("iterations", args.iterations)
    sr.set("temporalAreaRadius", r)
    sr.set("outputMemory", L.SRX_MEM_HOST if args.host_io else L.SRX_MEM_DEVICE)
    sr.set("readAhead", 1 if read_ahead else 0)
    if args.flow == "tvl1" and not R.dry:
        sr.setOpticalFlow(api.DualTVL1())
    elif args.flow == "farneback" and not R.dry:
        sr.setOpticalFlow(api.Farneback())
    else:
        counter = {"i": 0}

        def flow_cb(_f0, _f1):
            f = flows[counter["i"]]
            counter["i"] += 1
            return f.numpy() if R.dry else f

        sr.setOpticalFlow(flow_cb)
    if args.host_io or R.dry:
        left, own, right = ([f.cpu().numpy() for f in v] for v in (left, own, right))
    spans.configure_span(sr, left, own, right)

    def step():
        out = sr.nextFrame()
        if out is None:
            raise RuntimeError("stream ended inside the benchmark")
        return out

    R.sync()
    t_span0 = time.perf_counter()
    step()  # ring warm-up: 2r+2 reads and the first solves (not a steady-state step)
    for _ in range(args.warmup):
        step()
    timing = not args.no_kernel_timing and not R.dry
    sr.set("kernelTiming", 1 if timing else 0)
    R.sync()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        step_ms.append((time.perf_counter() - ts) * 1e3)
    if not R.dry:
        sr.synchronize()
    R.sync()
    t1 = time.perf_counter()
    elapsed, span_elapsed = t1 - t0, t1 - t_span0
    fwd_ms, bwd_ms, n_it = sr.kernelTimes() if timing else (0.0, 0.0, 0)
    sr.set("kernelTiming", 0)
    elapsed = R.max_over_ranks(elapsed)
    span_elapsed = R.max_over_ranks(span_elapsed)

    if rank == 0:
        P = args.rows * s * args.cols * s
        K = 2 * r + 1
        mpix = P / 1e6
        bytes_fwd, bytes_bwd = algorithmic_bytes(P, args.channels, K, s)
        headline_shape = args.config_name == "configs[3]"
        span_outputs = 1 + args.warmup + args.steps
        line = {
            "metric": METRIC,
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
                "workload": f"{args.config_name}: {args.cols}x{args.rows} {'RGB' if args.channels == 3 else 'gray'} -> "
                            f"{args.cols * s}x{args.rows * s}, {s}x BTV-L1, {args.iterations} iterations, "
                            f"temporalAreaRadius {r} (window {K}), streaming nextFrame(), motion: "
                            + ("exact synthetic flow (dry run)" if R.dry else
                               {"tvl1": "built-in TV-L1 flow", "farneback": "built-in Farneback flow"}.get(
                                   args.flow, "exact synthetic flow resident in HBM"))
                            + (" (stress clip: rotation amplitude growing along the clip)" if args.growing_rotation else ""),
                "frames_per_gpu": args.steps,
                "io": "host frames in, host frames out (PCIe inclusive)" if args.host_io else "frames resident in HBM",
                "read_ahead": read_ahead,
                "partition": "contiguous span of output frames per GPU, halo of r LR frames swapped point to point",
                "halo_exchange_ms": round(halo_ms, 3),
                "span_inclusive": {
                    "value": world * span_outputs * mpix / span_elapsed, "unit": "Mpix/s",
                    "ms": round(span_elapsed * 1e3, 2), "outputs_per_gpu": span_outputs,
                    "what": "every rank's first call (ring warm-up; r+2 solves on the rank that holds the start of the "
                            "stream) and the warm-up steps inside the clock, max over ranks"},
                "step_ms": [round(v, 2) for v in step_ms],
                "delivered_frame": f"{args.cols * s - 14}x{args.rows * s - 14} (cropped by btvKernelSize)",
            },
        }
        if R.rehearse:
            line["config"]["rehearsal"] = "dry-run handles on CPU over gloo" if R.dry else "all ranks on GPU 0 over gloo"
        if timing:
            bwd_launch_s = max(bwd_ms / max(n_it, 1) / 1e3, 1e-12)
            fwd_launch_s = max(fwd_ms / max(n_it, 1) / 1e3, 1e-12)
            kernels = {}
            # which kernels the library picks for this scale (default parameters otherwise)
            fwd_name = {4: "k_forward_sign_lds", 2: "k_forward_sign_s2w" if (args.cols * s) % 4 == 0 else "k_forward_sign_s2"}.get(
                s, "k_forward_sign_tile")
            bwd_name = {4: "k_backproject_update_s4", 2: "k_backproject_update_s4"}.get(s, "k_backproject_update")
            for name, what, nbytes, secs in (
                    (fwd_name, "sign(y_k - D H M_k X) of all window frames", bytes_fwd, fwd_launch_s),
                    (bwd_name, "back-projection of all window frames + BTV + update", bytes_bwd, bwd_launch_s)):
                tag = {"configs[3]": "", "configs[4]": "cfg5"}.get(args.config_name)
                traffic, traffic_src = measured_traffic(name, tag) if tag is not None else (None, None)
                kernels[name] = {"kernel": f"{name} ({what})", "achieved": nbytes / secs / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": nbytes / secs / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                                 "traffic_source": traffic_src, "algorithmic_bytes_per_launch": nbytes,
                                 "avg_launch_us": secs * 1e6}
            # the dominant kernel is the one the solver spends more time in
            dominant = fwd_name if fwd_launch_s >= bwd_launch_s else bwd_name
            other = bwd_name if dominant == fwd_name else fwd_name
            line["roofline"] = dict(
                {"bound": "hbm"}, **kernels[dominant],
                other_kernel=kernels[other],
                iteration={"algorithmic_bytes": bytes_fwd + bytes_bwd,
                           "achieved": (bytes_fwd + bytes_bwd) / (fwd_launch_s + bwd_launch_s) / 1e9,
                           "frac": (bytes_fwd + bytes_bwd) / (fwd_launch_s + bwd_launch_s) / 1e9 / HBM_PEAK_GBS},
                step={"algorithmic_bytes": (bytes_fwd + bytes_bwd) * args.iterations,
                      "frac": (bytes_fwd + bytes_bwd) * args.iterations / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                      "what": "solver bytes of one output frame over the whole step time (flow, map building, "
                              "crop and launch gaps included)"})
            ceil = copy_ceilings(dev)
            line["roofline"]["copy_ceiling"] = {
                "value": round(ceil["own_float4_kernel"], 1), "unit": "GB/s",
                "what": "device-to-device copy of 1 GiB by the library's own float4 copy kernel, read + write bytes",
                "torch_copy": round(ceil["torch_copy"], 1),
                "frac_of_ceiling": line["roofline"]["achieved"] / ceil["own_float4_kernel"]}
        if world == 1 and not args.no_cpu_baseline and not R.dry:
            base = min(r, len(frames) - 1)
            win = frames[:2 * r + 1]
            line["cpu_baseline"] = cpu_baseline(args, win, flows[:2 * r], base)
        print(json.dumps(line), flush=True)
    R.finish()


def _new_handle(args, R, flows_from=None):
    """A handle set up as the benchmark uses it; flows_from(i) gives the exact motion frame i -> i + 1 of the
    handle's stream for --flow synthetic and for dry-run handles."""
    from superresolution_amd import api, _lib as L

    sr = api.BTV_L1(-1 if R.dry else R.local_rank)
    sr.set("scale", args.scale)
    sr.set("iterations", args.iterations)
    sr.set("temporalAreaRadius", args.radius)
    sr.set("outputMemory", L.SRX_MEM_HOST if args.host_io else L.SRX_MEM_DEVICE)
    read_ahead = args.flow != "synthetic" and not args.no_read_ahead and not R.dry
    sr.set("readAhead", 1 if read_ahead else 0)
    if args.flow == "tvl1" and not R.dry:
        sr.setOpticalFlow(api.DualTVL1())
    elif args.flow == "farneback" and not R.dry:
        sr.setOpticalFlow(api.Farneback())
    else:
        counter = {"i": 0}

        def flow_cb(_f0, _f1):
            f = flows_from(counter["i"])
            counter["i"] += 1
            return f.numpy() if R.dry else f

        sr.setOpticalFlow(flow_cb)
    return sr, read_ahead


def bench_strong(args):
    """Strong scaling: one clip of `frames` outputs, one contiguous span of outputs per rank."""
    R = Ranks(args)
    torch = R.torch
    from superresolution_amd import spans, synth

    r, s = args.radius, args.scale
    world, rank, dev = R.world, R.rank, R.dev
    need = world * (2 * r + 1)  # srx_span_plan wants every rank to be fed 2r+1 frames at least
    total = args.frames if args.frames > 0 else max((args.steps + args.warmup) * world, need)
    if total < need:
        print(f"bench.py: --frames {total} is too short for {world} spans of radius {r} (need {need})", file=sys.stderr)
        sys.exit(2)
    a, b, in_first, in_last = spans.span_plan(total, r, world, rank)
    n_own = b - a + 1
    cams = synth.camera_path(total, seed=1234, rot_growth=args.growing_rotation)

    def frame(t):
        hr = synth.ground_truth_frame(cams[t], args.rows * s, args.cols * s, args.channels, 1234, dev)
        return synth.degrade(hr, s)

    def exact_flow(t):
        return synth.true_forward_flow(cams[t], cams[t + 1], args.rows, args.cols, s, dev)

    def stream_span(left, own, right, first):
        """All outputs of one span: (seconds, outputs)."""
        sr, read_ahead = _new_handle(args, R, lambda i: exact_flow(first + i))
        if args.host_io or R.dry:
            left, own, right = ([f.cpu().numpy() for f in v] for v in (left, own, right))
        spans.configure_span(sr, left, own, right)
        if not R.dry:
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        n_out = 0
        while sr.nextFrame() is not None:
            n_out += 1
        if not R.dry:
            sr.synchronize()
        return time.perf_counter() - t0, n_out, read_ahead

    # ---- optional: the whole clip on rank 0 alone (T1), the other ranks idle at the barrier
    single_s = args.single_ms / 1e3
    if args.with_single and world > 1:
        if rank == 0:
            whole = [frame(t) for t in range(total)]
            single_s, n_single, _ = stream_span([], whole, [], 0)
            assert n_single == total
            del whole
        R.sync()

    # ---- the span's frames; the boundary frames first, so that they travel while the rest is made
    R.sync()
    t_h0 = time.perf_counter()
    own = [None] * n_own
    edge = [i for i in range(n_own) if i < r or i >= n_own - r] if world > 1 else []
    for i in edge:
        own[i] = frame(a + i)
    ex = None
    if world > 1:
        # HaloExchange reads the first and the last r entries of the list only; the others are made below
        edges = [f if f is not None else own[0] for f in own]
        if R.rehearse and not R.dry:
            edges = [f.cpu() for f in edges]
        ex = spans.HaloExchange(edges, r, rank, world)
    for i in range(n_own):
        if own[i] is None:
            own[i] = frame(a + i)
    if not R.dry:
        torch.cuda.synchronize(dev)
    t_h1 = time.perf_counter()
    left, right = ex.wait() if ex else ([], [])
    if not R.dry:
        torch.cuda.synchronize(dev)
    exposed_ms = (time.perf_counter() - t_h1) * 1e3
    if R.rehearse and not R.dry:
        left, right = [f.to(dev) for f in left], [f.to(dev) for f in right]
    assert len(left) == a - in_first and len(right) == in_last - b
    halo_total_ms = (time.perf_counter() - t_h0) * 1e3

    # ---- timed region: inputs resident, every rank streams its span from its first call to its last output
    R.sync()
    t0 = time.perf_counter()
    own_s, n_out, read_ahead = stream_span(left, own, right, a - len(left))
    R.sync()
    elapsed = R.max_over_ranks(time.perf_counter() - t0)
    assert n_out == n_own, (n_out, n_own)
    slowest = R.max_over_ranks(own_s)
    fastest = -R.max_over_ranks(-own_s)
    exposed_max = R.max_over_ranks(exposed_ms)

    if rank == 0:
        P = args.rows * s * args.cols * s
        K = 2 * r + 1
        mpix = P / 1e6
        line = {
            "metric": METRIC,
            "value": total * mpix / elapsed,
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": total,
            "warmup": 0,
            "ms_per_step": elapsed / total * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config_name}: {args.cols}x{args.rows} {'RGB' if args.channels == 3 else 'gray'} -> "
                            f"{args.cols * s}x{args.rows * s}, {s}x BTV-L1, {args.iterations} iterations, "
                            f"temporalAreaRadius {r} (window {K}), one clip of {total} output frames, motion: "
                            + ("exact synthetic flow (dry run)" if R.dry else
                               {"tvl1": "built-in TV-L1 flow", "farneback": "built-in Farneback flow"}.get(
                                   args.flow, "exact synthetic flow resident in HBM")),
                "frames": total,
                "partition": "contiguous span of output frames per GPU (srx_span_plan), halo of r LR frames swapped "
                             "point to point; every rank's ring warm-up and the clip's first and last windows are "
                             "inside the clock",
                "frames_per_gpu": {"rank0": n_own, "min": total // world, "max": -(-total // world)},
                "io": "host frames in, host frames out (PCIe inclusive)" if args.host_io else "frames resident in HBM",
                "read_ahead": read_ahead,
                "span_seconds": {"slowest": round(slowest, 4), "fastest": round(fastest, 4)},
                "halo": {"total_ms": round(halo_total_ms, 3), "exposed_ms": round(exposed_max, 3),
                         "what": "boundary frames are synthesised first and posted, the rest of the span is synthesised "
                                 "while they travel; exposed = the wait in front of the first call, max over ranks "
                                 "(outside `value`, whose clock starts with every input resident in HBM)",
                         "value_with_exposed_halo": total * mpix / (elapsed + exposed_max / 1e3)},
            },
        }
        if single_s > 0:
            line["config"]["single_gpu_ms"] = round(single_s * 1e3, 2)
            line["efficiency"] = single_s / (world * elapsed)
        if R.rehearse:
            line["config"]["rehearsal"] = "dry-run handles on CPU over gloo" if R.dry else "all ranks on GPU 0 over gloo"
        print(json.dumps(line), flush=True)
    R.finish()


def bench_tvl1(args):
    """The flow stage alone: pairs of consecutive grey frames of the synthetic clip, sharded over the ranks.  A step
    is one library call on a batch of independent pairs, --pairs-in-flight of them side by side."""
    R = Ranks(args)
    torch = R.torch
    if R.dry:
        print("bench.py: the tvl1 workload needs a GPU", file=sys.stderr)
        sys.exit(2)
    import numpy as np
    from superresolution_amd import api, spans, synth, _lib as L

    rows, cols, dev = args.rows, args.cols, R.dev
    in_flight = max(1, args.pairs_in_flight)
    batch = args.pairs_per_step if args.pairs_per_step > 0 else 2 * in_flight
    n_pairs = (args.warmup + args.steps) * batch
    # pair t = frames (t, t + 1) of the clip; ranks own disjoint contiguous ranges of pairs
    first, last = spans.pair_shard(R.world * n_pairs, R.world, R.rank)
    assert last - first == n_pairs
    cams = synth.camera_path(R.world * n_pairs + 1, seed=4321)
    gray = [synth.ground_truth_frame(cams[first + t], rows, cols, 1, 4321, dev)[:, :, 0].to(torch.float32).contiguous()
            for t in range(n_pairs + 1)]
    sr = api.BTV_L1(R.local_rank)
    sr.setOpticalFlow(api.DualTVL1())
    lib = L.lib()
    import ctypes as C

    def step(k, lanes=in_flight):
        t = k * batch
        _flows, its = sr.calcFlowDevicePairs(gray[t:t + batch], gray[t + 1:t + batch + 1], in_flight=lanes)
        w = C.c_longlong()
        L.check(lib.srx_flow_pixel_iterations(sr.handle, C.byref(w)))
        return sum(its) / batch, w.value / batch

    for k in range(args.warmup):
        step(k)
    R.sync()
    t0 = time.perf_counter()
    iters, work, step_ms = [], [], []
    for k in range(args.warmup, args.warmup + args.steps):
        ts = time.perf_counter()
        i, w = step(k)
        step_ms.append((time.perf_counter() - ts) * 1e3)
        iters.append(i)
        work.append(w)
    R.sync()
    elapsed = R.max_over_ranks(time.perf_counter() - t0)
    # the same pairs one at a time (outside the clock): what a caller with a single pair in hand gets
    one_ms = None
    if in_flight > 1:
        step(args.warmup, lanes=1)
        ts = time.perf_counter()
        for k in range(args.warmup, args.warmup + min(args.steps, 3)):
            step(k, lanes=1)
        one_ms = (time.perf_counter() - ts) * 1e3 / (min(args.steps, 3) * batch)
    if R.rank == 0:
        bytes_pair = 64.0 * float(np.mean(work))  # 16 floats read or written per pixel and inner iteration
        # HBM-side bytes per pair of all flow kernels from the committed PMC passes (scripts/pmc_tvl1.sh), 1080p only
        pair_traffic = pair_traffic_src = None
        if (rows, cols) == (1080, 1920):
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_tvl1_traffic.json")))
            try:
                if files:
                    pair_traffic = json.load(open(files[-1]))["traffic_bytes_per_pair"]
                    pair_traffic_src = os.path.relpath(files[-1], ROOT)
            except (OSError, ValueError, KeyError):
                pass
        per_pair_s = elapsed / (args.steps * batch)
        line = {
            "metric": f"TV-L1 flow stage, {cols}x{rows} grey frame pairs per second",
            "value": R.world * args.steps * batch / elapsed,
            "unit": "pairs/s",
            "n_gpus": R.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"configs[2]: TV-L1 dual optical flow standalone on {cols}x{rows} grey frame pairs, 5 scales x "
                            f"5 warps x 300 inner iterations at most (stops per level and warp at epsilon 0.01); a step "
                            f"is one call on {batch} independent pair{'s' if batch > 1 else ''}, {in_flight} in flight "
                            f"side by side",
                "partition": "frame pairs sharded over the ranks, no exchange",
                "pairs_per_step": batch,
                "pairs_in_flight": in_flight,
                "ms_per_pair": per_pair_s * 1e3,
                "one_pair_at_a_time_ms_per_pair": None if one_ms is None else round(one_ms, 3),
                "pairs_per_gpu": {f"rank{g}": list(spans.pair_shard(R.world * n_pairs, R.world, g))
                                  for g in range(R.world)},
                "executed_inner_iterations_per_pair": float(np.mean(iters)),
                "pixel_iterations_per_pair": float(np.mean(work)),
                "step_ms": [round(v, 2) for v in step_ms],
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_tvl1_rows on levels of more than 200 000 pixels (2 or 4 inner iterations per "
                                           "launch), k_tvl1_multi below (8 inner iterations per launch)",
                "achieved": bytes_pair / per_pair_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": bytes_pair / per_pair_s / 1e9 / HBM_PEAK_GBS, "traffic": pair_traffic,
                "traffic_source": pair_traffic_src, "algorithmic_bytes_per_pair": bytes_pair,
                "what": "64 B x executed pixel-iterations over the whole time per pair (pyramid, warps, gradient passes "
                        "and the host's looks at the convergence words included)"},
        }
        if R.world == 1 and not args.no_cpu_baseline:
            import oracle.tvl1 as ot

            # one pair of the workload at its full size through the scalar C restatement (a few seconds); its flow
            # and its executed iteration count must be the GPU's
            a, b = (g.cpu().numpy() for g in gray[:2])
            tc = time.perf_counter()
            ref_flow, n_it = ot.calc_gray(a, b)
            dt = time.perf_counter() - tc
            got_flows, got_its = sr.calcFlowDevicePairs(gray[0:2], gray[1:3], in_flight=min(in_flight, 2))
            same = bool(np.array_equal(ref_flow, got_flows[0].cpu().numpy())) and int(got_its[0]) == int(n_it)
            if not same:
                raise RuntimeError("cpu_baseline: the oracle's flow differs from the GPU's for the same pair")
            line["cpu_baseline"] = {
                "value": 1.0 / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
                "sample": f"one {cols}x{rows} pair of the workload ({dt:.2f} s, {n_it} inner iterations, scalar C "
                          f"restatement)",
                "equals_gpu_output": same}
        print(json.dumps(line), flush=True)
    R.finish()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if args.workload == "tvl1":
        bench_tvl1(args)
    elif args.scaling == "strong":
        bench_strong(args)
    else:
        bench_solver(args)


if __name__ == "__main__":
    main()
