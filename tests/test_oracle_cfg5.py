"""CPU tests of the oracle's config-5 chain (deconvolution: fp/convolution.cpp:351-546).
Parity unpinned (no reference fixtures); pins are mathematical: FFT round trip, spectral division
recovers a known IR when smoothing is off, smoothing only changes magnitudes smoothly, shifteroo
swaps halves, invertFilter (*) h ~ delta."""
import numpy as np
import pytest

import oracle
from irbaboon_amd.synth import decaying_ir, exp_sine_sweep, white_noise


@pytest.mark.parametrize("n", [0, 1, 2, 3, 255, 256, 257, 65536, 65537])
def test_next_power_of_two(n):
    p = oracle.next_pow2(n)
    if n > 0 and (n & (n - 1)) == 0:
        assert p == n
    else:
        assert p > n and (p & (p - 1)) == 0 and p // 2 <= max(n, 1)


@pytest.mark.parametrize("n", [100, 1024, 5000, 1 << 16])
def test_fft_transform_roundtrip_and_layout(n):
    x = white_noise(n, n)
    F = oracle.fft_transform(x)
    N = oracle.next_pow2(n)
    assert F.size == 2 * N
    ref = np.fft.rfft(np.pad(x.astype(np.float64), (0, N - n)))
    got = F[0:N + 2:2] + 1j * F[1:N + 2:2]
    assert np.abs(got - ref).max() < 1e-5 * np.sqrt(N)
    y = oracle.fft_inv_transform(F)
    assert y.size == N and np.abs(y[:n] - x).max() < 5e-6


def test_shifteroo():
    assert np.array_equal(oracle.shifteroo(np.arange(6, dtype=np.float32)), [3, 4, 5, 0, 1, 2])
    assert np.array_equal(oracle.shifteroo(np.arange(5, dtype=np.float32)), [3, 4, 0, 1, 2])


def test_deconvolve_recovers_ir_without_smoothing():
    # the plug-in's geometry: 49152-sample sweep + 16384 silence = 65536 (Source/PluginProcessor.h:137-139)
    sr = 48000
    sweep = exp_sine_sweep(49152, sr, 20.0, sr / 2, fade_from=(11.0 / 12.0) * sr / 2, n_total=65536)
    h = decaying_ir(77, 512)
    cap = np.convolve(sweep.astype(np.float64), h.astype(np.float64))[:65536].astype(np.float32)
    got = oracle.deconvolve(cap, sweep, sr, smoothing=False)
    assert got.size == 65536
    # the sweep has no energy below 20 Hz and little near Nyquist, so compare band-limited
    G, Href = np.fft.rfft(got.astype(np.float64)), np.fft.rfft(np.pad(h.astype(np.float64), (0, 65536 - 512)))
    band = slice(int(40 / sr * 65536), int(20000 / sr * 65536))
    assert np.sqrt(np.mean(np.abs(G[band] - Href[band]) ** 2) / np.mean(np.abs(Href[band]) ** 2)) < 1e-3


def test_deconvolve_well_conditioned_exact():
    den = decaying_ir(5, 4096)
    den[0] += 4.0                                  # |D(f)| bounded away from zero
    h = decaying_ir(6, 300)
    num = np.fft.irfft(np.fft.rfft(den.astype(np.float64)) * np.fft.rfft(np.pad(h.astype(np.float64), (0, 4096 - 300)))).astype(np.float32)
    got = oracle.deconvolve(num, den, 48000, smoothing=False)
    assert np.sqrt(np.mean((got[:300] - h) ** 2) / np.mean(h ** 2)) < 1e-5
    assert np.abs(got[300:]).max() < 1e-5


def test_averaging_filter_properties():
    N = 4096
    x = decaying_ir(9, N)
    F = oracle.fft_transform(x)
    S = oracle.averaging_filter(F, 1.0 / 13.0, 48000.0)
    a0 = np.hypot(F[0:N + 2:2], F[1:N + 2:2])
    a1 = np.hypot(S[0:N + 2:2], S[1:N + 2:2])
    # phases kept, magnitudes smoothed (log-average inside the window, so within the local min/max)
    ph0, ph1 = np.arctan2(F[1:N + 2:2], F[0:N + 2:2]), np.arctan2(S[1:N + 2:2], S[0:N + 2:2])
    assert np.abs(np.angle(np.exp(1j * (ph0 - ph1))))[1:N // 2].max() < 1e-3
    for b in (50, 200, 1000):
        lo, hi = int(round(b / 2 ** (1 / 26))), int(round(b * 2 ** (1 / 26)))
        assert a0[lo:hi + 1].min() * 0.999 <= a1[b] <= a0[lo:hi + 1].max() * 1.001
        assert a1[b] == pytest.approx(np.exp(np.mean(np.log(a0[lo:hi + 1]))), rel=2e-3)
    # a flat spectrum is a fixed point
    flat = np.zeros(2 * N, np.float32)
    flat[0:N + 2:2] = 2.0
    out = oracle.averaging_filter(flat, 1.0 / 13.0, 48000.0)
    assert np.allclose(out[0:N // 2:2], 2.0, rtol=1e-4)
    # amplitude / phase switches (fp/convolution.cpp:537-538)
    Sa = oracle.averaging_filter(F, 1.0 / 13.0, 48000.0, include_phase=False)
    assert np.abs(Sa[1:N + 2:2]).max() < 1e-6 and np.all(Sa[0:N + 2:2] >= 0)
    Sp = oracle.averaging_filter(F, 1.0 / 13.0, 48000.0, include_amplitude=False)
    assert np.allclose(np.hypot(Sp[0:N + 2:2], Sp[1:N + 2:2])[1:N // 2], 1.0, atol=1e-5)


def test_invert_filter_roundtrip():
    # SURVEY.md section 4, KAT 3: invertFilter(h) (*) h ~ delta (circular); default smoothing on
    h = decaying_ir(3, 2048)
    h[0] += 2.0
    inv = oracle.invert_filter(h, 48000)
    c = np.fft.irfft(np.fft.rfft(h.astype(np.float64)) * np.fft.rfft(inv.astype(np.float64)))
    assert abs(c[0]) > 0.8 and np.abs(c[1:]).max() < 0.35 * abs(c[0])


def test_fft_transform_ampl_phase_quirk():
    """fp/tools.cpp:337-342: the amplitude is written over the real part before binPhase reads the pair, so the oracle's
    phase is atan2(im, amplitude) -- never outside [-pi/2, pi/2] -- not the bin's true phase."""
    x = white_noise(77, 1000)
    F, P = oracle.fft_transform(x), oracle.fft_transform(x, ampl_phase=True)
    N = oracle.next_pow2(1000)
    re, im = F[0:N + 2:2].astype(np.float64), F[1:N + 2:2].astype(np.float64)
    amp = np.sqrt(re ** 2 + im ** 2)
    assert np.allclose(P[0:N + 2:2], amp, rtol=1e-6)
    assert np.allclose(P[1:N + 2:2], np.arctan2(im, amp), atol=1e-6)
    assert np.abs(P[1:N + 2:2]).max() <= np.pi / 2 + 1e-6 and (re < 0).any()          # true phases would exceed pi/2 where re < 0
    assert np.array_equal(P[N + 2:], F[N + 2:])                                        # the loop stops at bin N/2 (`bin <= N`)
