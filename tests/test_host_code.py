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


# ---- independent checks: float64 formulations written from the mathematics, not from the reference's text ----
# (the bit-exact tests above compare two restatements of the same reference lines: a shared misreading would pass)

def _windows(nbins, fraction):
    ratio = 2.0 ** (fraction / 2.0)
    c = np.arange(nbins, dtype=np.float64)
    return np.floor(c / ratio + 0.5).astype(np.int64), np.floor(c * ratio + 0.5).astype(np.int64)


def _smoothed_amplitude_f64(amp, n, fraction):
    """geometric mean of |X| over [c / 2^(f/2), c * 2^(f/2)] (whole bins), the spectrum counted as 1e-16 between
    Nyquist and bin n-1 and as absent (but still in the window length) beyond: exact, via double prefix sums"""
    nbins = amp.size
    lo, hi = _windows(nbins, fraction)
    logs = np.concatenate([np.log(np.maximum(amp.astype(np.float64), 1e-16)), np.full(n - nbins, np.log(1e-16))])
    S = np.concatenate([[0.0], np.cumsum(logs)])
    return np.exp((S[np.minimum(hi, n - 1) + 1] - S[lo]) / (hi - lo + 1.0))


@pytest.mark.parametrize("n", [1 << 10, 1 << 14, 1 << 16])
def test_averaging_filter_against_float64_definition(n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) * np.exp(-np.arange(n) / (n / 6.0))).astype(np.float32)
    X = np.fft.rfft(x.astype(np.float64))
    buf = np.zeros(2 * n, np.float32)
    buf[0:n + 2:2], buf[1:n + 2:2] = X.real, X.imag
    Xf = buf[0:n + 2:2].astype(np.float64) + 1j * buf[1:n + 2:2].astype(np.float64)      # the float32 spectrum the filter sees
    out = buf.copy()
    ib.averagingFilter(out, 1.0 / 13.0, 48000.0)
    Y = out[0:n + 2:2].astype(np.float64) + 1j * out[1:n + 2:2].astype(np.float64)
    want = _smoothed_amplitude_f64(np.abs(Xf), n, 1.0 / 13.0)
    # amplitude: the product keeps the reference's single sliding float accumulator, whose rounding history drifts
    # from the exact mean by ~1e-4 relative at these sizes (test_sliding_float_sum_drift) -- far below a smoothing error
    assert np.abs(np.abs(Y) / want - 1.0).max() < 1e-3
    lo, hi = _windows(n // 2 + 1, 1.0 / 13.0)
    one = hi == lo                                   # windows of a single bin: nothing to average, amplitude unchanged
    assert one.sum() > 5 and np.abs(np.abs(Y[one]) / np.abs(Xf[one]) - 1.0).max() < 1e-6
    # phase is kept
    assert np.abs(np.angle(Y * np.conj(Xf))).max() < 1e-5
    assert not out[n + 2:].any()
    # include flags: amplitude only / phase only
    a = buf.copy()
    ib.averagingFilter(a, 1.0 / 13.0, 48000.0, includePhase=False)
    A = a[0:n + 2:2].astype(np.float64) + 1j * a[1:n + 2:2]
    assert np.abs(A.imag).max() == 0.0 and np.abs(A.real / want - 1.0).max() < 1e-3
    p = buf.copy()
    ib.averagingFilter(p, 1.0 / 13.0, 48000.0, includeAmplitude=False)
    Pp = p[0:n + 2:2].astype(np.float64) + 1j * p[1:n + 2:2]
    assert np.abs(np.abs(Pp) - 1.0).max() < 1e-6 and np.abs(np.angle(Pp * np.conj(Xf))).max() < 1e-5


def test_averaging_filter_flat_spectrum_and_high_frequency_rolloff():
    """|X| = 1 everywhere: the geometric mean is 1 wherever the window stays below Nyquist; where it runs past it
    the spectrum counts as 1e-16 and the documented roll-off appears (fp/convolution.hpp:40-43)."""
    n = 1 << 12
    buf = np.zeros(2 * n, np.float32)
    buf[0:n + 2:2] = 1.0
    ib.averagingFilter(buf, 1.0 / 3.0, 48000.0)
    amp = np.abs(buf[0:n + 2:2].astype(np.float64) + 1j * buf[1:n + 2:2])
    lo, hi = _windows(n // 2 + 1, 1.0 / 3.0)
    inside = hi <= n // 2
    assert np.abs(amp[inside] - 1.0).max() < 1e-5
    assert np.all(np.diff(amp[~inside]) <= 1e-9) and amp[-1] < 1e-1          # monotone roll-off towards Nyquist
    assert np.abs(amp / _smoothed_amplitude_f64(np.ones(n // 2 + 1), n, 1.0 / 3.0) - 1.0).max() < 1e-4


def test_averaging_filter_linear_mean_against_float64():
    n = 1 << 12
    rng = np.random.default_rng(3)
    buf = np.zeros(2 * n, np.float32)
    buf[0:n + 2] = rng.standard_normal(n + 2).astype(np.float32)
    amp = np.abs(buf[0:n + 2:2].astype(np.float64) + 1j * buf[1:n + 2:2])
    out = buf.copy()
    ib.averagingFilter(out, 1.0 / 6.0, 44100.0, logAvg=False)
    lo, hi = _windows(n // 2 + 1, 1.0 / 6.0)
    padded = np.concatenate([amp, np.zeros(n - amp.size)])
    S = np.concatenate([[0.0], np.cumsum(padded)])
    want = (S[np.minimum(hi, n - 1) + 1] - S[lo]) / (hi - lo + 1.0)            # arithmetic mean, zero above Nyquist
    got = np.abs(out[0:n + 2:2].astype(np.float64) + 1j * out[1:n + 2:2])
    assert np.abs(got / want - 1.0).max() < 1e-5


def test_sliding_float_sum_drift():
    """Why the smoothing sum stays a sequential host loop: the reference's ONE float accumulator (entering bins added,
    leaving bins subtracted) has a rounding history; the exact mean differs from it by far more than the 1e-5 parity
    budget, so a parallel (prefix-sum) formulation could not reproduce the reference's samples."""
    n = 1 << 16
    rng = np.random.default_rng(5)
    buf = np.zeros(2 * n, np.float32)
    X = np.exp(rng.normal(-6.0, 2.0, n // 2 + 1)) * np.exp(1j * rng.uniform(-3, 3, n // 2 + 1))
    buf[0:n + 2:2], buf[1:n + 2:2] = X.real, X.imag
    amp = np.abs(buf[0:n + 2:2].astype(np.float64) + 1j * buf[1:n + 2:2])
    out = buf.copy()
    ib.averagingFilter(out, 1.0 / 13.0, 48000.0)
    got = np.abs(out[0:n + 2:2].astype(np.float64) + 1j * out[1:n + 2:2])
    drift = np.abs(got / _smoothed_amplitude_f64(amp, n, 1.0 / 13.0) - 1.0)
    assert 2e-5 < drift.max() < 1e-2


def test_exp_sine_sweep_against_float64_formulas():
    """Farina sweep written from the formulas: x[i] = g sin(K (e^(i/L) - 1)), L = T / ln(w2/w1), K = w1 L; the inverse
    filter is the time-reversed sweep with a -6 dB/octave envelope; frequency doubles every T / log2(f2/f1) samples."""
    sr, dur, f1, f2, gain_db = 48000.0, 0.75, 30.0, 18000.0, -3.0
    T = sr * dur
    w1, w2 = 2 * np.pi * f1 / sr, 2 * np.pi * f2 / sr
    L = T / np.log(w2 / w1)
    i = np.arange(int(T), dtype=np.float64)
    want = 10 ** (gain_db / 20) * np.sin(w1 * L * (np.exp(i / L) - 1.0))
    s = ib.ExpSineSweep()
    s.generate(dur, sr, f1, f2, gain_db)
    assert s.getSweep().size == int(T) and np.abs(s.getSweep() - want).max() < 1e-9
    # instantaneous frequency of the phase K (e^(i/L) - 1) is w1 e^(i/L): starts at f1, reaches f2 at i = T
    assert w1 * np.exp(T / L) == pytest.approx(w2, rel=1e-12)
    s.generateInv()
    k = 10 ** (-6.0 * np.log2(f2 / f1) / 20.0 / T)
    assert np.abs(s.sweepInv - want[::-1] * k ** (i + 1)).max() < 1e-9
    octaves = np.log2(f2 / f1)
    for f in (31.0, 100.0, 1000.0, 9999.0, 17990.0):
        idx = int(np.floor(np.log2(f / f1) / octaves * T + 0.5))
        assert s.getSampleIndexAtFreq(f) == idx
        assert s.getFreqAtSampleIndex(idx) == pytest.approx(f1 * 2 ** (idx / T * octaves), rel=1e-12)
        assert abs(s.getFreqAtSampleIndex(idx) / f - 1.0) < 2.0 / T * octaves          # within one sample's frequency step
    assert s.getSampleIndexAtFreq(29.0) == -1 and s.getSampleIndexAtFreq(18001.0) == -1
    assert s.getSampleIndexAtFreq(17999.9) == -1          # rounds to sample T itself, which the sweep does not have
    at = s.getSampleIndexAtFreq(12000.0)
    before = s.getSweep().copy()
    s.linFadeout(12000.0)
    ramp = np.ones(int(T))
    ramp[at:] = (int(T) - at - np.arange(int(T) - at)) / (int(T) - at)
    assert np.abs(s.getSweep() - before * ramp).max() < 1e-12
