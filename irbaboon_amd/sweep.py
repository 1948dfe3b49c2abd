"""Host-side mirror of fp::ExpSineSweep (fp/ExpSineSweep.{hpp,cpp}): the Farina exponential sine sweep that
feeds the deconvolution chain.  Host code in the library (double precision, like the reference)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

_dp = C.POINTER(C.c_double)


class ExpSineSweep:
    def __init__(self):
        self._par = None
        self.sweep = np.zeros(0, np.float64)
        self.sweepInv = np.zeros(0, np.float64)

    def generate(self, durationSecs, sampleRate, lowFreq, highFreq, dBGain):
        self._par = (float(durationSecs), float(sampleRate), float(lowFreq), float(highFreq))
        out = np.zeros(int(sampleRate * durationSecs) + 1, np.float64)
        n = _lib.lib().irb_sweep_generate(*self._par, float(dBGain), out.ctypes.data_as(_dp), out.size)
        if n < 0:
            check(n)
        self.sweep = out[:n].copy()
        self._gain = float(dBGain)

    def generateInv(self):
        out = np.zeros(self.sweep.size + 1, np.float64)
        n = _lib.lib().irb_sweep_generate_inv(*self._par, self._gain, out.ctypes.data_as(_dp), out.size)
        if n < 0:
            check(n)
        self.sweepInv = out[:n].copy()

    def getSweep(self):
        return self.sweep

    def getSweepFloat(self):
        return self.sweep.astype(np.float32)

    def getSampleIndexAtFreq(self, freq):
        return _lib.lib().irb_sweep_sample_index_at_freq(float(freq), *self._par)

    def getFreqAtSampleIndex(self, index):
        return _lib.lib().irb_sweep_freq_at_sample_index(int(index), *self._par)

    def linFadeout(self, freq):
        idx = self.getSampleIndexAtFreq(freq)
        check(_lib.lib().irb_sweep_lin_fadeout(self.sweep.ctypes.data_as(_dp), self.sweep.size, idx))
