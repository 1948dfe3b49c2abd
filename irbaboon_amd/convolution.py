"""Host-side mirror of the reference's fp::convolution namespace for the hot path
(fp/convolution.hpp:18-32): same names, argument meaning and error behaviour, running on
the MI355X through the C ABI."""
import enum
import sys

import numpy as np

from . import _lib
from ._lib import planar_ptrs


class ChannelLayout(enum.IntEnum):          # fp/convolution.hpp:18-24
    unknown = 0
    IRMonoAudioMono = 1
    IRMonoAudioStereo = 2
    IRStereoAudioMono = 3
    IRStereoAudioStereo = 4


def classify(num_channels_ir, num_channels_audio):   # fp/convolution.cpp:28-37
    if num_channels_ir == 1 and num_channels_audio == 1:
        return ChannelLayout.IRMonoAudioMono
    if num_channels_ir == 1 and num_channels_audio == 2:
        return ChannelLayout.IRMonoAudioStereo
    if num_channels_ir == 2 and num_channels_audio == 1:
        return ChannelLayout.IRStereoAudioMono
    if num_channels_ir == 2 and num_channels_audio == 2:
        return ChannelLayout.IRStereoAudioStereo
    return ChannelLayout.unknown


def convolvePeriodic(buffer1, buffer2, processBlockSize=256, device=0, keep_reference_tail=True):
    """fp::convolution::convolvePeriodic(buffer1 = audio, buffer2 = IR, processBlockSize)
    (fp/convolution.cpp:14-242).  Planar float32 arrays [channels][samples]; returns
    [audio channels][La + Lh - 1].  An unknown channel layout gives a cleared buffer plus a
    debug line on stderr, as the reference does (:39-42).  keep_reference_tail=True keeps the
    reference's dropped tail (:236-238); False returns the complete linear convolution."""
    a = np.ascontiguousarray(np.atleast_2d(buffer1), np.float32)
    h = np.ascontiguousarray(np.atleast_2d(buffer2), np.float32)
    # (every sample of `out` is written by the call -- computed, or cleared: the dropped tail, the layout error, empty inputs)
    out = np.empty((a.shape[0], max(a.shape[1] + h.shape[1] - 1, 0)), np.float32)
    rc = _lib.lib().irb_convolve_periodic(device, planar_ptrs(a), a.shape[0], a.shape[1],
                                          planar_ptrs(h), h.shape[0], h.shape[1], processBlockSize,
                                          1 if keep_reference_tail else 0, planar_ptrs(out))
    if rc == _lib.IRB_ERR_LAYOUT:
        print("Either buffer1 or buffer2 is not mono nor stereo. Abort abort", file=sys.stderr)
        return out
    _lib.check(rc)
    return out


def convolveNonPeriodic(buffer1, buffer2, device=0):
    """fp::convolution::convolveNonPeriodic(buffer1 = audio, buffer2 = IR) (fp/convolution.cpp:246-347):
    one large FFT per signal on the GPU.  Unknown layout: cleared buffer + debug line (:271-275)."""
    a = np.ascontiguousarray(np.atleast_2d(buffer1), np.float32)
    h = np.ascontiguousarray(np.atleast_2d(buffer2), np.float32)
    # (every sample of `out` is written by the call -- computed, or cleared: the layout error, empty inputs)
    out = np.empty((a.shape[0], max(a.shape[1] + h.shape[1] - 1, 0)), np.float32)
    rc = _lib.lib().irb_convolve_nonperiodic(device, planar_ptrs(a), a.shape[0], a.shape[1],
                                             planar_ptrs(h), h.shape[0], h.shape[1], planar_ptrs(out))
    if rc == _lib.IRB_ERR_LAYOUT:
        print("Either buffer1 or buffer2 is not mono nor stereo. Abort abort", file=sys.stderr)
        return out
    _lib.check(rc)
    return out


def IRtoRealFFTRaw(buffer, irPartSize, device=0):
    """fp::ir::IRtoRealFFTRaw (fp/ir.cpp:106-147): [len/irPartSize + 1][2*irPartSize] packed partition spectra."""
    import ctypes as C
    ir = np.ascontiguousarray(buffer, np.float32).reshape(-1)
    rows = ir.size // irPartSize + 1
    out = np.zeros((rows, 2 * irPartSize), np.float32)
    n = C.c_int()
    _lib.check(_lib.lib().irb_ir_to_real_fft_raw(device, ir.ctypes.data_as(_lib.fp), ir.size, irPartSize,
                                                 out.ctypes.data_as(_lib.fp), out.size, C.byref(n)))
    return out
