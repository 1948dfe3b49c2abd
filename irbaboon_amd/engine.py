"""FdlEngine: Python handle on the batched, device-resident FDL convolver (irb_fdl_*)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FdlConfig, FdlInfo, check, fp, planar_ptrs


class FdlEngine:
    """n_instances independent IR x stream instances of `channels` channels each, all with
    block size `block` and IRs up to `max_ir_len` samples, resident on one MI355X.

    Mirrors the state the reference keeps per plug-in instance (Source/PluginProcessor.h:194-217)
    -- FDL ring, IR partition spectra, overlap -- batched over instances."""

    def __init__(self, n_instances, channels, block, max_ir_len, max_blocks_per_call=1,
                 device=0, stream=None, flags=0):
        self._h = C.c_void_p()
        cfg = FdlConfig(device, n_instances, channels, block, max_ir_len, max_blocks_per_call,
                        C.c_void_p(stream) if stream else None, flags)
        check(_lib.lib().irb_fdl_create(C.byref(cfg), C.byref(self._h)))
        self.n_instances, self.channels, self.block = n_instances, channels, block
        self.streams = n_instances * channels
        self.max_blocks_per_call = max_blocks_per_call
        info = FdlInfo()
        check(_lib.lib().irb_fdl_get_info(self._h, C.byref(info)))
        self.info = info

    def close(self):
        try:
            if self._h:
                _lib.lib().irb_fdl_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:          # interpreter shutdown: the module may already be torn down
            pass

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_ir(self, instance, ir):
        """ir: float32 [ir_channels][len] (or 1-D for mono)."""
        ir = np.ascontiguousarray(np.atleast_2d(ir), np.float32)
        check(_lib.lib().irb_fdl_set_ir(self._h, instance, planar_ptrs(ir), ir.shape[0], ir.shape[1]))

    def set_ir_partition(self, instance, partition, samples):
        s = np.ascontiguousarray(samples, np.float32)
        assert s.size == self.block
        check(_lib.lib().irb_fdl_set_ir_partition(self._h, instance, partition, s.ctypes.data_as(fp)))

    def get_ir_spectra(self, instance, channel, n_partitions):
        """[n_partitions][N] float32: packed partition spectra (im(0) := re(N/2), fp/ir.cpp:123-124)."""
        out = np.empty((n_partitions, self.info.fft_len), np.float32)
        check(_lib.lib().irb_fdl_get_ir_spectra(self._h, instance, channel, out.ctypes.data_as(fp), n_partitions))
        return out

    def set_ir_spectra(self, instance, spectra):
        """spectra: float32 [ir_channels][n_partitions][N] (or [n_partitions][N] for mono)."""
        sp = np.ascontiguousarray(spectra, np.float32)
        if sp.ndim == 2:
            sp = sp[None]
        flat = sp.reshape(sp.shape[0], -1)
        check(_lib.lib().irb_fdl_set_ir_spectra(self._h, instance, planar_ptrs(flat), sp.shape[0], sp.shape[1]))

    def reset(self):
        check(_lib.lib().irb_fdl_reset(self._h))

    def process(self, x):
        """x: float32 [streams][n_blocks*block] (stream = instance*channels + channel); returns same shape."""
        x = np.ascontiguousarray(x, np.float32)
        assert x.ndim == 2 and x.shape[0] == self.streams and x.shape[1] % self.block == 0
        y = np.empty_like(x)
        check(_lib.lib().irb_fdl_process(self._h, planar_ptrs(x), planar_ptrs(y), x.shape[1] // self.block))
        return y

    def process_refresh(self, x, ir, first_partition, instance=0):
        """The real-time callback body (irb_fdl_process_refresh): block j of x re-derives partition (first_partition + j) mod P
        of `instance` from the time-domain `ir` (P*block samples) before it is stepped.  One wait for the whole call."""
        x = np.ascontiguousarray(x, np.float32)
        ir = np.ascontiguousarray(ir, np.float32)
        assert x.ndim == 2 and x.shape[0] == self.streams and x.shape[1] % self.block == 0
        assert ir.size >= self.info.partitions * self.block
        y = np.empty_like(x)
        check(_lib.lib().irb_fdl_process_refresh(self._h, instance, planar_ptrs(x), planar_ptrs(y), x.shape[1] // self.block,
                                                 ir.ctypes.data_as(fp), first_partition))
        return y

    def mac_kernel_name(self, n_blocks):
        """which kernel a call of n_blocks blocks per stream multiplies with on this handle (irb_fdl_mac_kernel_name)"""
        return _lib.lib().irb_fdl_mac_kernel_name(self._h, n_blocks).decode()

    def counters(self):
        v = [C.c_ulonglong() for _ in range(3)]
        check(_lib.lib().irb_fdl_get_counters(self._h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def process_packed(self, x, out=None):
        """x: one C-contiguous float32 array [streams][n_blocks*block]; no per-stream packing on the host.
        Use pinned_empty() buffers for PCIe-speed transfers."""
        assert x.dtype == np.float32 and x.flags.c_contiguous and x.shape[0] == self.streams and x.shape[1] % self.block == 0
        y = np.empty_like(x) if out is None else out
        check(_lib.lib().irb_fdl_process_packed(self._h, x.ctypes.data, y.ctypes.data, x.shape[1], x.shape[1] // self.block))
        return y

    # ---- device-resident path (bench, smoke) ----
    def device_io(self):
        d_in, d_out, stride = C.c_void_p(), C.c_void_p(), C.c_long()
        check(_lib.lib().irb_fdl_device_io(self._h, C.byref(d_in), C.byref(d_out), C.byref(stride)))
        return d_in.value, d_out.value, stride.value

    def upload(self, d_dst, src):
        src = np.ascontiguousarray(src, np.float32)
        check(_lib.lib().irb_fdl_upload(self._h, d_dst, src.ctypes.data_as(fp), src.size))

    def download(self, d_src, n):
        out = np.empty(n, np.float32)
        check(_lib.lib().irb_fdl_download(self._h, out.ctypes.data_as(fp), d_src, n))
        return out

    def process_device(self, d_in, d_out, stride, n_blocks):
        check(_lib.lib().irb_fdl_process_device(self._h, d_in, d_out, stride, n_blocks))

    def synchronize(self):
        check(_lib.lib().irb_fdl_synchronize(self._h))

    def mark(self, slot):
        check(_lib.lib().irb_fdl_mark(self._h, slot))

    def elapsed_ms(self):
        ms = C.c_float()
        check(_lib.lib().irb_fdl_elapsed_ms(self._h, C.byref(ms)))
        return ms.value


class MultiGpuEngine:
    """Instances sharded by contiguous ranges over several devices of ONE process (irb_fdl_multi_*)."""

    def __init__(self, n_instances, channels, block, max_ir_len, devices, max_blocks_per_call=1, flags=0):
        self._h = C.c_void_p()
        cfg = FdlConfig(0, n_instances, channels, block, max_ir_len, max_blocks_per_call, None, flags)
        dev = (C.c_int * len(devices))(*devices)
        check(_lib.lib().irb_fdl_multi_create(C.byref(cfg), dev, len(devices), C.byref(self._h)))
        self.streams, self.block = n_instances * channels, block

    def close(self):
        try:
            if self._h:
                _lib.lib().irb_fdl_multi_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    __del__ = close

    def owner(self, instance):
        d, li = C.c_int(), C.c_int()
        check(_lib.lib().irb_fdl_multi_owner(self._h, instance, C.byref(d), C.byref(li)))
        return d.value, li.value

    def set_ir(self, instance, ir):
        ir = np.ascontiguousarray(np.atleast_2d(ir), np.float32)
        check(_lib.lib().irb_fdl_multi_set_ir(self._h, instance, planar_ptrs(ir), ir.shape[0], ir.shape[1]))

    def reset(self):
        check(_lib.lib().irb_fdl_multi_reset(self._h))

    def process(self, x):
        x = np.ascontiguousarray(x, np.float32)
        assert x.ndim == 2 and x.shape[0] == self.streams and x.shape[1] % self.block == 0
        y = np.empty_like(x)
        check(_lib.lib().irb_fdl_multi_process(self._h, planar_ptrs(x), planar_ptrs(y), x.shape[1] // self.block))
        return y


class _PinnedBlock:
    """Owns one irb_host_alloc block and exposes it through the buffer protocol of a ctypes array; numpy arrays
    built on it keep it alive through their .base chain."""

    def __init__(self, n_floats):
        self.ptr = _lib.lib().irb_host_alloc(max(n_floats, 1) * 4)
        if not self.ptr:
            raise _lib.IrbError(_lib.IRB_ERR_HIP, _lib.lib().irb_last_error().decode())
        self.array = (C.c_float * max(n_floats, 1)).from_address(self.ptr)
        self.array._owner = self                       # ctypes objects accept attributes: ties lifetimes together

    def __del__(self):
        try:
            _lib.lib().irb_host_free(self.ptr)
        except Exception:
            pass


def pinned_empty(shape):
    """float32 numpy array backed by page-locked host memory (irb_host_alloc) for the packed API."""
    n = int(np.prod(shape))
    block = _PinnedBlock(n)
    return np.frombuffer(block.array, dtype=np.float32, count=n).reshape(shape)
