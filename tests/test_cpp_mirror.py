"""Builds and runs the C++ host-mirror test (tests/cpp/test_host_mirror.cpp): the reference is
compiled C++, so its drop-in callers are C++ too."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")


def _build(hip_lib):
    so_dir = os.path.join(ROOT, "irbaboon_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", EXE, os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp"),
                           "-L" + so_dir, "-l:libirbfdl.so", "-Wl,-rpath," + so_dir, "-Wl,-rpath,/opt/rocm/lib"])


def test_cpp_mirror_layout_error_convention(hip_lib):
    _build(hip_lib)
    r = subprocess.run([EXE, "layout"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "not mono nor stereo" in r.stderr
    # the adapter whose create failed printed exactly one line, whatever the number of callbacks that followed
    assert r.stderr.count("irb_rt:") == 1, r.stderr


@pytest.mark.gpu
def test_cpp_mirror_delta_delay_on_gpu(hip_lib):
    _build(hip_lib)
    r = subprocess.run([EXE, "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
