"""CPU checks of the drop-in boundary: libirbfdl.so loads, exports every symbol
include/irb_fdl.h declares, and refuses to compute without a GPU (no silent fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import irbaboon_amd
from irbaboon_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "irb_fdl.h")).read() + open(os.path.join(ROOT, "include", "irb_deconv.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(irb_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(hip_lib):
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/irb_fdl.h but not exported"
    assert {s[0] for s in _lib.SYMBOLS} == set(names)


def test_config_struct_layout_matches_header():
    assert C.sizeof(_lib.FdlConfig) == 40 and C.sizeof(_lib.FdlInfo) == 40


def _no_gpu():
    return not os.path.exists("/dev/kfd")


@pytest.mark.skipif(not _no_gpu(), reason="GPU present")
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(irbaboon_amd.IrbError) as e:
        irbaboon_amd.FdlEngine(1, 1, 256, 1024)
    assert e.value.code == _lib.IRB_ERR_HIP
    with pytest.raises(irbaboon_amd.IrbError):
        irbaboon_amd.convolvePeriodic(np.ones((1, 512), np.float32), np.ones((1, 64), np.float32), 64)


def test_layout_error_convention():
    # unknown layout: cleared output and no exception, as fp/convolution.cpp:39-42 (checked before any HIP call)
    y = irbaboon_amd.convolvePeriodic(np.ones((3, 64), np.float32), np.ones((1, 16), np.float32), 16)
    assert y.shape == (3, 79) and not y.any()


def test_argument_errors():
    l = _lib.lib()
    assert l.irb_fdl_create(None, None) == _lib.IRB_ERR_ARG
    assert b"null" in l.irb_last_error()
    cfg = _lib.FdlConfig(0, 0, 1, 256, 100, 1, None, 0)
    h = C.c_void_p()
    assert l.irb_fdl_create(C.byref(cfg), C.byref(h)) == _lib.IRB_ERR_ARG
    p = C.c_void_p()
    assert l.irb_rt_create(0, 2, 256, 2048, C.byref(p)) == 0          # host-only state, no GPU needed
    assert l.irb_rt_partitions(p) == 8 and l.irb_rt_latency(p) == 256  # Source/PluginProcessor.cpp:59,225
    buf = np.zeros((2, 64), np.float32)
    assert l.irb_rt_process_block(p, _lib.planar_ptrs(buf), 64) == _lib.IRB_ERR_STATE
    l.irb_rt_destroy(p)
