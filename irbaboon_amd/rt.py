"""RtConvolver: the convolution part of IRBaboonAudioProcessor (prepareToPlay / processBlock /
releaseResources, Source/PluginProcessor.cpp:164-248, 408-562) on the GPU through irb_rt_*."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, fp, planar_ptrs


class RtConvolver:
    def __init__(self, channels=2, block=256, ir_ring_len=2048, device=0):
        self._h = C.c_void_p()
        check(_lib.lib().irb_rt_create(device, channels, block, ir_ring_len, C.byref(self._h)))
        self.channels = channels

    def close(self):
        try:
            if self._h:
                _lib.lib().irb_rt_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    __del__ = close

    def prepareToPlay(self, samplesPerBlock):
        check(_lib.lib().irb_rt_prepare_to_play(self._h, samplesPerBlock))

    def releaseResources(self):
        check(_lib.lib().irb_rt_release_resources(self._h))

    def setIR(self, ir):
        ir = np.ascontiguousarray(ir, np.float32)
        check(_lib.lib().irb_rt_set_ir(self._h, ir.ctypes.data_as(fp), ir.size))

    def processBlock(self, buffer):
        """buffer: float32 C-contiguous [channels][n], processed in place."""
        assert buffer.dtype == np.float32 and buffer.flags.c_contiguous and buffer.shape[0] == self.channels
        check(_lib.lib().irb_rt_process_block(self._h, planar_ptrs(buffer), buffer.shape[1]))

    def processBlockBypassed(self, buffer):
        """Source/PluginProcessor.cpp:584-590: keeps the block delay while the plug-in is bypassed."""
        assert buffer.dtype == np.float32 and buffer.flags.c_contiguous and buffer.shape[0] == self.channels
        check(_lib.lib().irb_rt_process_block_bypassed(self._h, planar_ptrs(buffer), buffer.shape[1]))

    def time_callbacks(self, n, n_callbacks):
        """Wall time in microseconds of each of n_callbacks processBlock calls of n samples per channel on white noise,
        timed in C on the calling thread (irb_rt_time_callbacks)."""
        us = np.empty(n_callbacks, np.float32)
        check(_lib.lib().irb_rt_time_callbacks(self._h, n, n_callbacks, us.ctypes.data_as(fp)))
        return us

    def counters(self):
        """(callbacks, kernel launches, blocking waits, copy commands) since prepareToPlay."""
        v = [C.c_ulonglong() for _ in range(4)]
        check(_lib.lib().irb_rt_counters(self._h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    @property
    def latency(self):
        return _lib.lib().irb_rt_latency(self._h)

    @property
    def partitions(self):
        return _lib.lib().irb_rt_partitions(self._h)
