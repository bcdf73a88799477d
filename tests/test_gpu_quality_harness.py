"""Runs the C++ quality harness (samples/quality_harness.cpp: the measurement loop of the reference's
samples/quality_test/main.cpp:269-358 on the C++ facade) on a small synthetic clip.  The program is
built beforehand by __graft_entry__.build() (samples/Makefile)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "samples", "bin", "quality_harness")


@pytest.mark.gpu
@pytest.mark.parametrize("flow", ["farneback", "tvl1"])
@pytest.mark.parametrize("channels", [1, 3])
def test_super_resolution_beats_bicubic(channels, flow):
    assert os.path.exists(EXE), "samples/bin/quality_harness is missing: run __graft_entry__.build()"
    r = subprocess.run([EXE, "--rows", "128", "--cols", "160", "--frames", "9", "--temporal", "2", "--iterations",
                        "60", "--channels", str(channels), "--opt-flow", flow], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = re.findall(r"\|\s+\[\s*(\d+)\]\s+\|\s+([\d.]+) / ([\d.]+)\s+\|\s+([\d.]+) / ([\d.]+)", r.stdout)
    assert "Optical Flow      : " + flow in r.stdout
    assert len(rows) == 8                     # one output per input frame after the skipped first one (main.cpp:273-287)
    sr = sum(float(x[1]) for x in rows) / len(rows)
    bc = sum(float(x[3]) for x in rows) / len(rows)
    assert sr > bc + 1.0, r.stdout            # at least 1 dB over bicubic on this clip


@pytest.mark.gpu
def test_clip_whose_size_is_not_a_multiple_of_the_scale_is_cut_as_the_reference_cuts_it():
    """AllignedFrameSource (samples/quality_test/main.cpp:67-75): 131 x 163 at scale 4 runs as 128 x 160 and gives
    the same table as the clip generated at 128 x 160."""
    args = ["--frames", "7", "--temporal", "1", "--iterations", "20", "--channels", "3"]
    a = subprocess.run([EXE, "--rows", "131", "--cols", "163"] + args, capture_output=True, text=True, timeout=100)
    b = subprocess.run([EXE, "--rows", "128", "--cols", "160"] + args, capture_output=True, text=True, timeout=100)
    assert a.returncode == 0 and b.returncode == 0, a.stdout + a.stderr + b.stdout + b.stderr
    assert "synthetic 163x131 (used: 160x128)" in a.stdout
    table = lambda out: [ln for ln in out.splitlines() if ln.startswith("|  [") or "Avg" in ln or "AVG" in ln]
    assert len(table(a.stdout)) == 6 + 4 and table(a.stdout) == table(b.stdout)


def test_host_side_image_routines_of_the_harness():
    """The harness restates cv::GaussianBlur and cv::resize(INTER_CUBIC) on 8-bit data in fixed point
    (samples/quality_test/main.cpp:130-131, 318); its --selftest compares them with double-precision statements
    of the same sums: the blur exactly (rounded half up), the enlargement exactly given the 11-bit coefficients,
    a constant image within one grey level; and the counterpart of AllignedFrameSource (main.cpp:67-75) on a clip
    whose size is not a multiple of the scale.  No GPU involved."""
    if not os.path.exists(EXE):  # built by __graft_entry__.build(); a tree that only ran `make` in csrc/ builds it here
        subprocess.run(["make", "-C", os.path.join(ROOT, "samples")], check=True, capture_output=True, timeout=300)
    r = subprocess.run([EXE, "--selftest"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "selftest: ok" in r.stdout, r.stdout + r.stderr
