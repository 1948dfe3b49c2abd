"""Host-resident product code (no GPU needed): the pieces of the library that the north star keeps on the CPU
-- averagingFilter, shifteroo, nextPowerOfTwo, ExpSineSweep -- against the oracle's restatement (bit exact: same
libm, same operation order) and against their mathematical definition."""
import numpy as np
import pytest

import irbaboon_amd as ib
import oracle
from irbaboon_amd.synth import decaying_ir, exp_sine_sweep


@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 255, 256, 257, 65536, 65537, (1 << 20) + 1])
def test_next_power_of_two(n):
    assert ib.nextPowerOfTwo(n) == oracle.next_pow2(n)


def test_averaging_filter_and_shifteroo_bit_exact():
    N = 8192
    F = oracle.fft_transform(decaying_ir(9, N))
    for kw in ({}, {"includePhase": False}, {"includeAmplitude": False}, {"logAvg": False}):
        a = F.copy()
        ib.averagingFilter(a, 1.0 / 13.0, 48000.0, **kw)
        b = oracle.averaging_filter(F, 1.0 / 13.0, 48000.0, log_avg=kw.get("logAvg", True),
                                    include_phase=kw.get("includePhase", True), include_amplitude=kw.get("includeAmplitude", True))
        assert np.array_equal(a, b)
    for n in (1, 2, 5, 6, 1001):
        v = np.arange(n, dtype=np.float32)
        w = v.copy()
        ib.shifteroo(w)
        assert np.array_equal(w, oracle.shifteroo(v))


# the plug-in's own sweep: (sweepLength+1)/sampleRate seconds, 20 Hz .. Nyquist, fade out from 11/12 Nyquist
# (Source/PluginProcessor.cpp:28-30, Source/PluginProcessor.h:137-139)
SR, NSW = 48000.0, 49152
DUR = (np.float32(NSW) + 1) / np.float32(SR)


def test_exp_sine_sweep_matches_oracle_bit_exact():
    s = ib.ExpSineSweep()
    s.generate(float(DUR), SR, 20.0, SR / 2.0, 0.0)
    ref = oracle.sweep_generate(float(DUR), SR, 20.0, SR / 2.0)
    assert s.getSweep().size == ref.size == int(SR * float(DUR)) and np.array_equal(s.getSweep(), ref)
    f = (11.0 / 12.0) * SR / 2.0
    idx = s.getSampleIndexAtFreq(f)
    assert idx == oracle.sweep_index_at_freq(f, float(DUR), SR, 20.0, SR / 2.0) and 0 < idx < ref.size
    assert s.getFreqAtSampleIndex(idx) == oracle.sweep_freq_at_index(idx, float(DUR), SR, 20.0, SR / 2.0)
    s.linFadeout(f)
    assert np.array_equal(s.getSweep(), oracle.sweep_lin_fadeout(ref, idx))
    assert s.getSweep()[-1] != 0.0 and abs(s.getSweep()[-1]) < 1e-3          # fades to (almost) zero
    s.generateInv()
    assert np.array_equal(s.sweepInv, oracle.sweep_generate(float(DUR), SR, 20.0, SR / 2.0, inverse=True))
    assert s.getSampleIndexAtFreq(10.0) == -1 and s.getSampleIndexAtFreq(SR) == -1   # out of the sweep's range
    # the numpy generator used for synthetic inputs is the same formula (with T = n samples exactly)
    t = ib.ExpSineSweep()
    t.generate(49152 / SR, SR, 20.0, SR / 2.0, 0.0)
    t.linFadeout(f)
    mine = exp_sine_sweep(49152, SR, 20.0, SR / 2.0, fade_from=f)
    assert np.abs(mine - t.getSweepFloat()).max() < 2e-6


def test_exp_sine_sweep_properties():
    """Farina sweep: instantaneous frequency grows exponentially from lowFreq to highFreq; sweep (*) inverse sweep
    is a band-limited impulse."""
    s = ib.ExpSineSweep()
    s.generate(1.0, SR, 50.0, 12000.0, -6.0)
    x = s.getSweep()
    assert x.size == 48000 and np.abs(x).max() == pytest.approx(10 ** (-6 / 20), rel=1e-4)
    for frac in (0.1, 0.5, 0.9):
        i = int(frac * x.size)
        seg = x[i - 400:i + 400]                                    # zero-crossing rate = local frequency
        zc = np.sum(np.signbit(seg[:-1]) != np.signbit(seg[1:]))
        f_est = zc / 2 / (seg.size / SR)
        assert f_est == pytest.approx(s.getFreqAtSampleIndex(i), rel=0.08)
        assert s.getSampleIndexAtFreq(s.getFreqAtSampleIndex(i)) == i
    s.generateInv()
    c = np.convolve(x, s.sweepInv)
    pk = np.argmax(np.abs(c))
    assert abs(pk - (x.size - 1)) <= 2 and np.abs(c[pk]) > 8 * np.abs(c[:pk - 2000]).max()


def test_sweep_argument_errors():
    l = ib.lib()
    buf = np.zeros(10, np.float64)
    dp = buf.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double))
    assert l.irb_sweep_generate(1.0, 48000.0, 20.0, 24000.0, 0.0, dp, 10) < 0           # out too small
    assert l.irb_sweep_generate(1.0, 48000.0, 0.0, 24000.0, 0.0, dp, 10) < 0            # lowFreq must be > 0
    assert l.irb_sweep_sample_index_at_freq(100.0, 1.0, 48000.0, 200.0, 100.0) == -1   # high <= low
