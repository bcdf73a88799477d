"""Builds tests/cpp/facade_test.cpp against include/srx/superres.hpp + libsrx.so with g++ and runs it.
The program uses dry-run objects only, so it needs no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_facade(tmp_path):
    exe = str(tmp_path / "facade_test")
    libdir = os.path.join(ROOT, "superresolution_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"), "-o", exe, "-L", libdir, "-lsrx",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "facade ok" in r.stdout
