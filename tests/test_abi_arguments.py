"""Argument validation of the C ABI happens before any HIP call, so it is checkable without a GPU: every bad
call returns IRB_ERR_ARG (-1) / IRB_ERR_LAYOUT (-3) and leaves a message in irb_last_error()."""
import ctypes as C

import numpy as np
import pytest

from irbaboon_amd import _lib
from irbaboon_amd._lib import FdlConfig, fp, planar_ptrs

L = _lib.lib()


def _err():
    return L.irb_last_error().decode()


@pytest.mark.parametrize("cfg", [
    dict(n_instances=0), dict(channels=0), dict(block=0), dict(max_ir_len=0), dict(max_blocks_per_call=0),
    dict(block=4096),
])
def test_create_rejects_bad_geometry(cfg):
    base = dict(device=0, n_instances=1, channels=1, block=256, max_ir_len=1000, max_blocks_per_call=1)
    base.update(cfg)
    c = FdlConfig(base["device"], base["n_instances"], base["channels"], base["block"], base["max_ir_len"],
                  base["max_blocks_per_call"], None, 0)
    h = C.c_void_p()
    assert L.irb_fdl_create(C.byref(c), C.byref(h)) == _lib.IRB_ERR_ARG and not h.value and _err()


def test_null_handles_are_refused_everywhere():
    buf = np.zeros(8, np.float32)
    info = _lib.FdlInfo()
    assert L.irb_fdl_get_info(None, C.byref(info)) == _lib.IRB_ERR_ARG
    assert L.irb_fdl_reset(None) == _lib.IRB_ERR_ARG
    assert L.irb_fdl_synchronize(None) == _lib.IRB_ERR_ARG
    assert L.irb_fdl_process_device(None, None, None, 0, 0) == _lib.IRB_ERR_ARG
    assert L.irb_fdl_set_ir_partition(None, 0, 0, buf.ctypes.data_as(fp)) == _lib.IRB_ERR_ARG
    assert L.irb_rt_prepare_to_play(None, 256) == _lib.IRB_ERR_ARG
    assert L.irb_lfft_synchronize(None) == _lib.IRB_ERR_ARG
    assert L.irb_fdl_multi_reset(None) == _lib.IRB_ERR_ARG
    L.irb_fdl_destroy(None); L.irb_rt_destroy(None); L.irb_lfft_destroy(None); L.irb_fdl_multi_destroy(None)   # no-ops


def test_convolve_entry_points_validate_before_touching_the_gpu():
    a = np.ones((1, 64), np.float32)
    out = np.ones((1, 79), np.float32)
    h = np.ones((1, 16), np.float32)
    assert L.irb_convolve_periodic(0, planar_ptrs(a), 1, -1, planar_ptrs(h), 1, 16, 16, 1, planar_ptrs(out)) == _lib.IRB_ERR_ARG
    assert L.irb_convolve_periodic(0, planar_ptrs(a), 1, 64, planar_ptrs(h), 1, 16, 0, 1, planar_ptrs(out)) == _lib.IRB_ERR_ARG
    assert L.irb_convolve_periodic(0, None, 1, 64, planar_ptrs(h), 1, 16, 16, 1, planar_ptrs(out)) == _lib.IRB_ERR_ARG
    # unknown layout: the output is cleared first (reference fp/convolution.cpp:20-21,39-42)
    a3 = np.ones((3, 64), np.float32)
    out3 = np.ones((3, 79), np.float32)
    assert L.irb_convolve_periodic(0, planar_ptrs(a3), 3, 64, planar_ptrs(h), 1, 16, 16, 1, planar_ptrs(out3)) == _lib.IRB_ERR_LAYOUT
    assert not out3.any() and "mono nor stereo" in _err()
    out3[:] = 1
    assert L.irb_convolve_nonperiodic(0, planar_ptrs(a3), 3, 64, planar_ptrs(h), 1, 16, planar_ptrs(out3)) == _lib.IRB_ERR_LAYOUT
    assert not out3.any()
    # empty signals are answered without a device: all zeros
    out[:] = 1
    e = np.zeros((1, 0), np.float32)
    assert L.irb_convolve_periodic(0, planar_ptrs(e), 1, 0, planar_ptrs(h), 1, 16, 16, 1, planar_ptrs(out)) == _lib.IRB_OK
    assert not out[0, :15].any()


def test_config5_entry_points_validate_arguments():
    x = np.ones(100, np.float32)
    out = np.zeros(256, np.float32)
    assert L.irb_fft_transform(0, x.ctypes.data_as(fp), 0, out.ctypes.data_as(fp)) == _lib.IRB_ERR_ARG
    assert L.irb_fft_transform(0, x.ctypes.data_as(fp), 8, out.ctypes.data_as(fp)) == _lib.IRB_ERR_ARG      # < 17 samples
    assert L.irb_fft_inv_transform(0, x.ctypes.data_as(fp), 100, out.ctypes.data_as(fp)) == _lib.IRB_ERR_ARG  # not a power of two
    assert L.irb_deconvolve(0, x.ctypes.data_as(fp), 100, x.ctypes.data_as(fp), 0, 48000.0, 1, 1, 1, out.ctypes.data_as(fp), 256) == _lib.IRB_ERR_ARG
    assert L.irb_deconvolve(0, x.ctypes.data_as(fp), 100, x.ctypes.data_as(fp), 100, 48000.0, 1, 1, 1, out.ctypes.data_as(fp), 64) == _lib.IRB_ERR_ARG  # out too small
    p = C.c_void_p()
    for bad in (0, 24, 100, 1 << 24):
        assert L.irb_lfft_create(0, bad, 1, C.byref(p)) == _lib.IRB_ERR_ARG and not p.value
    assert L.irb_lfft_create(0, 1024, 0, C.byref(p)) == _lib.IRB_ERR_ARG
    n = C.c_int()
    raw = np.zeros(4096, np.float32)
    for bad_ps in (0, 8, 100, 4096):
        assert L.irb_ir_to_real_fft_raw(0, x.ctypes.data_as(fp), 100, bad_ps, raw.ctypes.data_as(fp), raw.size, C.byref(n)) == _lib.IRB_ERR_ARG
    assert L.irb_ir_to_real_fft_raw(0, x.ctypes.data_as(fp), 100, 64, raw.ctypes.data_as(fp), 10, C.byref(n)) == _lib.IRB_ERR_ARG   # out too small


def test_rt_adapter_state_machine_without_gpu():
    p = C.c_void_p()
    assert L.irb_rt_create(0, 0, 256, 2048, C.byref(p)) == _lib.IRB_ERR_ARG
    assert L.irb_rt_create(0, 2, 256, 2048, C.byref(p)) == _lib.IRB_OK
    short = np.zeros(100, np.float32)
    assert L.irb_rt_set_ir(p, short.ctypes.data_as(fp), 100) == _lib.IRB_ERR_ARG and "rolling refresh" in _err()   # reference :457 reads past it
    ok = np.zeros(2048, np.float32)
    assert L.irb_rt_set_ir(p, ok.ctypes.data_as(fp), 2048) == _lib.IRB_OK
    buf = np.zeros((2, 256), np.float32)
    assert L.irb_rt_process_block(p, planar_ptrs(buf), 256) == _lib.IRB_ERR_STATE
    assert L.irb_rt_process_block_bypassed(p, planar_ptrs(buf), 256) == _lib.IRB_ERR_STATE
    assert L.irb_rt_prepare_to_play(p, 0) == _lib.IRB_ERR_ARG
    assert L.irb_rt_release_resources(p) == _lib.IRB_OK
    L.irb_rt_destroy(p)


def test_multi_group_arguments():
    c = FdlConfig(0, 2, 1, 256, 1000, 1, None, 0)
    m = C.c_void_p()
    dev = (C.c_int * 3)(0, 0, 0)
    assert L.irb_fdl_multi_create(C.byref(c), dev, 3, C.byref(m)) == _lib.IRB_ERR_ARG      # more devices than instances
    assert L.irb_fdl_multi_create(C.byref(c), dev, 0, C.byref(m)) == _lib.IRB_ERR_ARG


def test_convolve_periodic_refuses_a_result_buffer_that_aliases_the_audio():
    """fp::convolution::convolvePeriodic returns a NEW AudioBuffer (fp/convolution.cpp:20-21): result and audio never share samples, and
    the kernels read the one while they write the other.  An overlapping `out` is an argument error, reported before any HIP call."""
    big = np.zeros(3000, np.float32)
    audio, out = big[100:1100][None], big[600:600 + 1000 + 50 - 1][None]          # La = 1000, Lh = 50: out overlaps the audio's second half
    h = np.ones((1, 50), np.float32)
    rc = L.irb_convolve_periodic(0, planar_ptrs(audio), 1, 1000, planar_ptrs(h), 1, 50, 16, 1, planar_ptrs(out))
    assert rc == _lib.IRB_ERR_ARG and "alias" in _err()
    p = _lib.ConvolveProfile()
    assert L.irb_device_quiet_begin(-1) == _lib.IRB_ERR_ARG and L.irb_device_quiet_end(0) == _lib.IRB_ERR_STATE       # unbalanced end, bad device
    assert L.irb_convolve_last_profile(None) == _lib.IRB_ERR_ARG
