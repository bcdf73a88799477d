"""bench.py as the driver calls it: `python bench.py --gpus N` starts the N ranks itself.  No GPU here, so the
ranks run dry-run handles over gloo (SRX_BENCH_REHEARSE=dry): the launcher, the rendezvous, the span plan, the
halo exchange, the per-rank call sequence and the JSON line are the real ones, the numbers mean nothing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, **env):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=e, capture_output=True,
                          text=True, timeout=600)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_and_reports_them():
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg1", "--iterations", "2"],
             SRX_BENCH_REHEARSE="dry")
    assert p.returncode == 0, p.stderr[-2000:]
    d = _json_line(p.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["halo_exchange_ms"] > 0
    assert d["config"]["span_inclusive"]["outputs_per_gpu"] == 4
    assert d["config"]["span_inclusive"]["value"] > 0 and d["value"] > 0
    assert "rehearsal" in d["config"]


def test_strong_scaling_cuts_one_clip_into_spans():
    """--scaling strong: one clip of --frames outputs, a span per rank; every output is delivered once, the line
    says `strong`, and with --with-single rank 0 streams the whole clip alone first (efficiency = T1 / (n Tn))."""
    p = _run(["--gpus", "3", "--scaling", "strong", "--frames", "31", "--workload", "cfg1", "--iterations", "2",
              "--with-single"], SRX_BENCH_REHEARSE="dry")
    assert p.returncode == 0, p.stderr[-2000:]
    d = _json_line(p.stdout)
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and d["steps"] == 31 and d["warmup"] == 0
    c = d["config"]
    assert c["frames"] == 31 and c["frames_per_gpu"]["min"] == 10 and c["frames_per_gpu"]["max"] == 11
    assert c["halo"]["exposed_ms"] >= 0 and c["halo"]["total_ms"] > 0 and c["halo"]["value_with_exposed_halo"] <= d["value"]
    assert c["single_gpu_ms"] > 0 and 0 < d["efficiency"]
    assert c["span_seconds"]["slowest"] >= c["span_seconds"]["fastest"] > 0
    # one rank: the same code path without an exchange
    p1 = _run(["--gpus", "1", "--scaling", "strong", "--frames", "12", "--workload", "cfg1", "--iterations", "2"],
              SRX_BENCH_REHEARSE="dry")
    assert p1.returncode == 0, p1.stderr[-2000:]
    d1 = _json_line(p1.stdout)
    assert d1["n_gpus"] == 1 and d1["scaling"] == "strong" and d1["config"]["halo"]["exposed_ms"] >= 0
    # a clip too short for the spans is refused by every rank before any of them waits for a neighbour
    p2 = _run(["--gpus", "2", "--scaling", "strong", "--frames", "10", "--workload", "cfg1"], SRX_BENCH_REHEARSE="dry")
    assert p2.returncode != 0 and "too short" in p2.stderr


def test_a_rank_count_other_than_asked_for_is_refused():
    p = _run(["--gpus", "1", "--workload", "cfg1"], SRX_BENCH_REHEARSE="dry", WORLD_SIZE="2", RANK="0")
    assert p.returncode != 0
    assert "refusing" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_a_failing_rank_fails_the_launcher():
    # the tvl1 workload cannot run on dry-run handles: the ranks exit non-zero and so must the parent
    p = _run(["--gpus", "2", "--workload", "tvl1"], SRX_BENCH_REHEARSE="dry")
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_measured_traffic_is_looked_up_per_workload():
    """The roofline objects quote HBM traffic from the committed counter passes: the headline workload from
    profiles/rNN_traffic.json, the scale-2 configuration from profiles/rNN_cfg5_traffic.json; a kernel or a
    workload without a pass has none."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t4, src4 = bench.measured_traffic("k_backproject_update_s4")
    t5, src5 = bench.measured_traffic("k_backproject_update_s4", "cfg5")
    assert t4 and src4.endswith("_traffic.json") and "cfg5" not in src4 and "tvl1" not in src4
    assert t5 and src5.endswith("_cfg5_traffic.json") and t5 != t4
    assert bench.measured_traffic("k_no_such_kernel", "cfg5") == (None, None)
    assert bench.measured_traffic("k_backproject_update_s4", "no_such_workload") == (None, None)
