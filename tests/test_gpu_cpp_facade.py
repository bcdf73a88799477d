"""The C++ facade on a GPU: tests/cpp/facade_stream_gpu.cpp streams the clip of tests/golden/stream_rgb_r2.npz
(made by the oracle, tests/golden/make_golden.py) through Algorithm::create("SuperResolution.BTV_L1") with a
one-output and a two-output flow plug-in, and through BTV_L1_Base::process, and compares bytes.  The program is
built beforehand by __graft_entry__.build() (samples/Makefile)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "samples", "bin", "facade_stream_gpu")


@pytest.mark.gpu
def test_cpp_facade_streams_the_golden_clip(tmp_path):
    assert os.path.exists(EXE), "samples/bin/facade_stream_gpu is missing: run __graft_entry__.build()"
    z = np.load(os.path.join(ROOT, "tests", "golden", "stream_rgb_r2.npz"))
    frames, flows, outs = z["frames"], z["flows"], z["outs"]
    n, rows, cols, ch = frames.shape
    np.ascontiguousarray(frames).tofile(tmp_path / "frames.bin")
    np.ascontiguousarray(flows, dtype=np.float32).tofile(tmp_path / "flows.bin")
    np.ascontiguousarray(outs).tofile(tmp_path / "outs.bin")
    (tmp_path / "meta.txt").write_text(
        f"{n} {rows} {cols} {ch} {int(z['radius'])} {int(z['iterations'])} {outs.shape[1]} {outs.shape[2]}\n")
    r = subprocess.run([EXE, str(tmp_path)], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "facade stream ok" in r.stdout
