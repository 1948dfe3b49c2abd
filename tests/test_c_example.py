"""The public headers are C (not only C++): they must compile as C11, and the plain-C example must build
against libirbfdl.so and, on a GPU, run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "batch_convolve")


def test_headers_compile_as_c11(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "irb_fdl.h"\n#include "irb_deconv.h"\nint main(void) { irb_fdl_config c = {0}; (void) c; return 0; }\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), str(src)])


def _build(hip_lib):
    so_dir = os.path.join(ROOT, "irbaboon_amd")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "batch_convolve.c"), "-o", EXE,
                           "-L" + so_dir, "-l:libirbfdl.so", "-Wl,-rpath," + so_dir, "-Wl,-rpath,/opt/rocm/lib", "-lm"])


def test_c_example_builds(hip_lib):
    _build(hip_lib)


@pytest.mark.gpu
def test_c_example_runs_on_gpu(hip_lib):
    _build(hip_lib)
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "delta-delay max err" in r.stdout
