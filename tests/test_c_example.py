"""The public headers are C (not only C++): they must compile as C11, and the plain-C example must build
against libirbfdl.so and, on a GPU, run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "batch_convolve")
EXE_RT = os.path.join(ROOT, "examples", "rt_callback")


def test_headers_compile_as_c11(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "irb_fdl.h"\n#include "irb_deconv.h"\nint main(void) { irb_fdl_config c = {0}; (void) c; return 0; }\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), str(src)])


def _build(hip_lib):
    so_dir = os.path.join(ROOT, "irbaboon_amd")
    for src, exe in (("batch_convolve.c", EXE), ("rt_callback.c", EXE_RT)):
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "examples", src), "-o", exe,
                               "-L" + so_dir, "-l:libirbfdl.so", "-Wl,-rpath," + so_dir, "-Wl,-rpath,/opt/rocm/lib", "-lm"])


def test_c_example_builds(hip_lib):
    _build(hip_lib)


@pytest.mark.gpu
def test_c_example_runs_on_gpu(hip_lib):
    _build(hip_lib)
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "delta-delay max err" in r.stdout


@pytest.mark.gpu
def test_c_rt_example_runs_on_gpu(hip_lib):
    """the plug-in callback driven from plain C: delta IR = pure delay of latency + 100 samples at host blocks 256 and 512;
    every callback is one wait and no launch (resident server) or, with IRB_RT_RESIDENT=0, one launch"""
    _build(hip_lib)
    for resident, want in (("1", "0.00 launches, 1.00 waits, 0.00 copies"), ("0", "1.00 launches, 1.00 waits, 0.00 copies")):
        r = subprocess.run([EXE_RT], capture_output=True, text=True, timeout=300, env=dict(os.environ, IRB_RT_RESIDENT=resident))
        assert r.returncode == 0, r.stdout + r.stderr
        lines = [l for l in r.stdout.splitlines() if l.startswith("host block")]
        assert len(lines) == 2 and all(want in l for l in lines), r.stdout
