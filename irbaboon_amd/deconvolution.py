"""Host-side mirror of the reference's config-5 chain: tools::fftTransform / fftInvTransform
(fp/tools.cpp:321-369), convolution::deconvolve / averagingFilter (fp/convolution.cpp:351-546),
ir::invertFilter / shifteroo (fp/ir.cpp:13-18, 85-103).  Same names and argument meaning; transforms and
the spectral division run on the MI355X through the C ABI of include/irb_deconv.h."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, fp


def _f(a):
    return a.ctypes.data_as(fp)


def nextPowerOfTwo(x):
    return _lib.lib().irb_next_power_of_two(int(x))


def fftTransform(buffer, formatAmplPhase=False, device=0):
    """mono float32 samples -> 2N floats, interleaved (re, im) for bins 0..N/2, N = nextPowerOfTwo(len);
    formatAmplPhase=True: (amplitude, phase) pairs as the reference forms them (fp/tools.cpp:337-342)."""
    x = np.ascontiguousarray(buffer, np.float32).reshape(-1)
    out = np.zeros(2 * nextPowerOfTwo(x.size), np.float32)
    fn = _lib.lib().irb_fft_transform_ampl_phase if formatAmplPhase else _lib.lib().irb_fft_transform
    check(fn(device, _f(x), x.size, _f(out)))
    return out


def fftInvTransform(buffer, device=0):
    b = np.ascontiguousarray(buffer, np.float32).reshape(-1)
    out = np.zeros(b.size // 2, np.float32)
    check(_lib.lib().irb_fft_inv_transform(device, _f(b), b.size, _f(out)))
    return out


def averagingFilter(buffer, octaveFraction, sampleRate, logAvg=True, includePhase=True, includeAmplitude=True):
    """In place on a float32 spectrum buffer (fftSize floats); host code, like the reference."""
    assert buffer.dtype == np.float32 and buffer.flags.c_contiguous
    _lib.lib().irb_averaging_filter(_f(buffer), buffer.size, float(octaveFraction), float(sampleRate),
                                    int(logAvg), int(includePhase), int(includeAmplitude))


def shifteroo(buffer):
    assert buffer.dtype == np.float32 and buffer.flags.c_contiguous
    _lib.lib().irb_shifteroo(_f(buffer), buffer.size)


def deconvolve(numeratorBuffer, denominatorBuffer, sampleRate, smoothing=True, includePhase=True,
               includeAmplitude=True, device=0):
    num = np.ascontiguousarray(numeratorBuffer, np.float32).reshape(-1)
    den = np.ascontiguousarray(denominatorBuffer, np.float32).reshape(-1)
    out = np.zeros(nextPowerOfTwo(max(num.size, den.size)), np.float32)
    check(_lib.lib().irb_deconvolve(device, _f(num), num.size, _f(den), den.size, float(sampleRate),
                                    int(smoothing), int(includePhase), int(includeAmplitude), _f(out), out.size))
    return out


def invertFilter(buffer, sampleRate, device=0):
    ir = np.ascontiguousarray(buffer, np.float32).reshape(-1)
    out = np.zeros(nextPowerOfTwo(ir.size), np.float32)
    check(_lib.lib().irb_invert_filter(device, _f(ir), ir.size, int(sampleRate), _f(out), out.size))
    return out


class LargeFft:
    """Reusable plan with device-resident buffers (irb_lfft_*)."""

    def __init__(self, fft_len, max_batch=1, device=0):
        self._h = C.c_void_p()
        check(_lib.lib().irb_lfft_create(device, fft_len, max_batch, C.byref(self._h)))
        self.N, self.max_batch = fft_len, max_batch
        t, s = C.c_void_p(), C.c_void_p()
        check(_lib.lib().irb_lfft_buffers(self._h, C.byref(t), C.byref(s)))
        self.d_time, self.d_spec = t.value, s.value

    def close(self):
        try:
            if self._h:
                _lib.lib().irb_lfft_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    __del__ = close

    def invalidate_spec(self):
        """after writing above the Nyquist bin of d_spec yourself: the next forward() restores the layout's zeros (irb_lfft_invalidate_spec)"""
        check(_lib.lib().irb_lfft_invalidate_spec(self._h))

    def forward(self, batch=1):
        check(_lib.lib().irb_lfft_forward_device(self._h, self.d_time, self.d_spec, batch))

    def inverse(self, batch=1):
        check(_lib.lib().irb_lfft_inverse_device(self._h, self.d_spec, self.d_time, batch))

    def synchronize(self):
        check(_lib.lib().irb_lfft_synchronize(self._h))

    def mark(self, slot):
        check(_lib.lib().irb_lfft_mark(self._h, slot))

    def elapsed_ms(self):
        ms = C.c_float()
        check(_lib.lib().irb_lfft_elapsed_ms(self._h, C.byref(ms)))
        return ms.value
