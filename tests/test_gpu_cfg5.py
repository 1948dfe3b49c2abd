"""GPU parity of the config-5 chain (large real FFT, spectral division, deconvolve, invertFilter) against
the oracle, through the C ABI.  The oracle for this chain is parity unpinned (no reference fixtures exist);
its own pins are in tests/test_oracle_cfg5.py."""
import numpy as np
import pytest

import oracle
import irbaboon_amd as ib
from irbaboon_amd.synth import decaying_ir, exp_sine_sweep, white_noise
from util import rel_rms

pytestmark = pytest.mark.gpu


def _bins(F, N):
    return F[0:N + 2:2].astype(np.float64) + 1j * F[1:N + 2:2].astype(np.float64)


@pytest.mark.parametrize("n", [17, 100, 1000, 4096, 5000, 1 << 13, 1 << 16, (1 << 16) + 1, 1 << 20])
def test_fft_transform_matches_oracle_and_numpy(n):
    x = white_noise(n, n)
    F = ib.fftTransform(x)
    N = ib.nextPowerOfTwo(n)
    assert F.size == 2 * N and N == oracle.next_pow2(n)
    ref = np.fft.rfft(np.pad(x.astype(np.float64), (0, N - n)))
    got = _bins(F, N)
    assert np.sqrt(np.mean(np.abs(got - ref) ** 2) / np.mean(np.abs(ref) ** 2)) < 2e-6
    assert rel_rms(np.abs(got), np.abs(_bins(oracle.fft_transform(x), N))) < 1e-5
    assert not F[N + 2:].any()                                  # unused upper half is cleared
    y = ib.fftInvTransform(F)
    assert y.size == N and rel_rms(y[:n], x) < 1e-5 and (n == N or np.abs(y[n:]).max() < 1e-5)


@pytest.mark.parametrize("n", [100, 4096, 1 << 16, 1 << 20])
def test_fft_transform_ampl_phase_form(n):
    """tools::fftTransform(buffer, formatAmplPhase=true), fp/tools.cpp:337-342: (amplitude, phase) per bin, with the
    reference's quirk that the phase is taken AFTER the amplitude replaced the real part (atan2(im, amplitude))."""
    x = white_noise(n + 3, n)
    F, Fc = ib.fftTransform(x, formatAmplPhase=True), ib.fftTransform(x)
    N = ib.nextPowerOfTwo(n)
    O = oracle.fft_transform(x, ampl_phase=True)
    assert rel_rms(F[0:N + 2:2], O[0:N + 2:2]) < 1e-5                       # amplitudes
    assert np.abs(F[1:N + 2:2] - O[1:N + 2:2]).max() < 2e-3                 # phases in [-pi/2, pi/2]: |im| <= amplitude
    amp = np.sqrt(Fc[0:N + 2:2].astype(np.float64) ** 2 + Fc[1:N + 2:2].astype(np.float64) ** 2)
    assert np.allclose(F[0:N + 2:2], amp, rtol=1e-6, atol=1e-6)
    assert np.allclose(F[1:N + 2:2], np.arctan2(Fc[1:N + 2:2].astype(np.float64), amp), atol=1e-6)
    assert not F[N + 2:].any()


def test_fft_transform_rejects_tiny_buffers():
    with pytest.raises(ib.IrbError):
        ib.fftTransform(np.ones(8, np.float32))


def test_host_helpers_match_oracle_exactly():
    N = 4096
    F = oracle.fft_transform(decaying_ir(9, N))
    for kw in ({}, {"includePhase": False}, {"includeAmplitude": False}, {"logAvg": False}):
        a = F.copy()
        ib.averagingFilter(a, 1.0 / 13.0, 48000.0, **kw)
        b = oracle.averaging_filter(F, 1.0 / 13.0, 48000.0, log_avg=kw.get("logAvg", True),
                                    include_phase=kw.get("includePhase", True), include_amplitude=kw.get("includeAmplitude", True))
        assert np.array_equal(a, b)                              # same libm, same operation order
    for n in (5, 6, 2, 1):
        v = np.arange(n, dtype=np.float32)
        w = v.copy()
        ib.shifteroo(w)
        assert np.array_equal(w, oracle.shifteroo(v))


@pytest.mark.parametrize("smoothing,phase,ampl", [(False, True, True), (True, True, True), (True, False, True), (True, True, False)])
def test_deconvolve_well_conditioned_matches_oracle(smoothing, phase, ampl):
    den = decaying_ir(5, 6000)
    den[0] += 4.0                                               # |D(f)| bounded away from zero
    num = np.convolve(den.astype(np.float64), decaying_ir(6, 300).astype(np.float64))[:6000].astype(np.float32)
    got = ib.deconvolve(num, den, 48000, smoothing, phase, ampl)
    ref = oracle.deconvolve(num, den, 48000, smoothing, phase, ampl)
    assert got.shape == ref.shape == (8192,)
    assert rel_rms(got, ref) < 1e-5


def test_deconvolve_plugin_geometry_sweep():
    """The plug-in's own call (Source/PluginProcessor.cpp:306): 65536-sample capture / sweep, defaults."""
    sr = 48000
    sweep = exp_sine_sweep(49152, sr, 20.0, sr / 2, fade_from=(11.0 / 12.0) * sr / 2, n_total=65536)
    h = decaying_ir(77, 512)
    cap = np.convolve(sweep.astype(np.float64), h.astype(np.float64))[:65536].astype(np.float32)
    for smoothing in (False, True):
        got = ib.deconvolve(cap, sweep, sr, smoothing)
        ref = oracle.deconvolve(cap, sweep, sr, smoothing)
        # the sweep has almost no energy below 20 Hz: those bins divide rounding noise by ~0 and differ
        # between any two FFT implementations, so compare where the division is conditioned (40 Hz..20 kHz)
        G, R = np.fft.rfft(got.astype(np.float64)), np.fft.rfft(ref.astype(np.float64))
        band = slice(int(40 / sr * 65536), int(20000 / sr * 65536))
        assert np.sqrt(np.mean(np.abs(G[band] - R[band]) ** 2) / np.mean(np.abs(R[band]) ** 2)) < 1e-4


def test_deconvolve_zero_denominator_bins_are_skipped():
    # den = delta - delayed delta has exact spectral zeros; fp/tools.cpp:74-77 leaves those bins alone
    N = 1024
    den = np.zeros(N, np.float32)
    den[0] = 1.0
    num = white_noise(3, N)
    assert rel_rms(ib.deconvolve(num, den, 48000, smoothing=False), num) < 1e-5      # division by 1
    z = np.zeros(N, np.float32)
    got = ib.deconvolve(num, z, 48000, smoothing=False)
    assert rel_rms(got, num) < 1e-5                                                  # every denominator 0: untouched
    assert rel_rms(got, oracle.deconvolve(num, z, 48000, smoothing=False)) < 1e-5


def test_invert_filter_matches_oracle():
    h = decaying_ir(3, 2048)
    h[0] += 2.0
    got, ref = ib.invertFilter(h, 48000), oracle.invert_filter(h, 48000)
    assert rel_rms(got, ref) < 1e-5
    c = np.fft.irfft(np.fft.rfft(h.astype(np.float64)) * np.fft.rfft(got.astype(np.float64)))
    assert abs(c[0]) > 0.8


def test_config5_full_size_one_million_samples():
    """BASELINE configs[4]: 2^20-sample sweep capture -> IR extraction; smoothing off so the answer is known."""
    sr, N = 48000, 1 << 20
    sweep = exp_sine_sweep(3 * N // 4, sr, 20.0, sr / 2, fade_from=(11.0 / 12.0) * sr / 2, n_total=N)
    h = decaying_ir(55, 4096)
    cap = np.fft.irfft(np.fft.rfft(sweep.astype(np.float64)) * np.fft.rfft(np.pad(h.astype(np.float64), (0, N - 4096)))).astype(np.float32)
    got = ib.deconvolve(cap, sweep, sr, smoothing=False)
    assert got.size == N
    G, Href = np.fft.rfft(got.astype(np.float64)), np.fft.rfft(np.pad(h.astype(np.float64), (0, N - 4096)))
    band = slice(int(40 / sr * N), int(20000 / sr * N))
    assert np.sqrt(np.mean(np.abs(G[band] - Href[band]) ** 2) / np.mean(np.abs(Href[band]) ** 2)) < 1e-3
    ref = oracle.deconvolve(cap, sweep, sr, smoothing=False)
    R = np.fft.rfft(ref.astype(np.float64))
    assert np.sqrt(np.mean(np.abs(G[band] - R[band]) ** 2) / np.mean(np.abs(R[band]) ** 2)) < 1e-4


def test_config5_full_size_default_smoothing_and_inversion():
    """BASELINE configs[4] at 2^20 samples with the chain's DEFAULTS: deconvolve with its three 1/13-octave
    smoothing passes (fp/convolution.cpp:389-394) and ir::invertFilter (fp/ir.cpp:13-18), against the oracle.
    Well-conditioned spectra (|D| bounded away from zero) so the whole band can be held to 1e-5; the sweep
    capture with smoothing is compared where the division is conditioned (40 Hz..20 kHz)."""
    sr, N = 48000, 1 << 20
    den = decaying_ir(5, N)
    den[0] += 4.0
    h = decaying_ir(6, 4096)
    num = np.fft.irfft(np.fft.rfft(den.astype(np.float64)) * np.fft.rfft(h.astype(np.float64), N), N).astype(np.float32)
    for phase, ampl in ((True, True), (False, True)):
        got = ib.deconvolve(num, den, sr, True, phase, ampl)
        ref = oracle.deconvolve(num, den, sr, True, phase, ampl)
        assert got.shape == ref.shape == (N,)
        assert rel_rms(got, ref) < 1e-5
    hh = decaying_ir(7, N)
    hh[0] += 2.0
    got, ref = ib.invertFilter(hh, sr), oracle.invert_filter(hh, sr)
    assert got.shape == ref.shape == (N,) and rel_rms(got, ref) < 1e-5
    sweep = exp_sine_sweep(3 * N // 4, sr, 20.0, sr / 2, fade_from=(11.0 / 12.0) * sr / 2, n_total=N)
    cap = np.fft.irfft(np.fft.rfft(sweep.astype(np.float64)) * np.fft.rfft(h.astype(np.float64), N), N).astype(np.float32)
    G = np.fft.rfft(ib.deconvolve(cap, sweep, sr, True).astype(np.float64))
    R = np.fft.rfft(oracle.deconvolve(cap, sweep, sr, True).astype(np.float64))
    band = slice(int(40 / sr * N), int(20000 / sr * N))
    assert np.sqrt(np.mean(np.abs(G[band] - R[band]) ** 2) / np.mean(np.abs(R[band]) ** 2)) < 1e-4


@pytest.mark.parametrize("fuse", ["0", "3"])
@pytest.mark.parametrize("tile", ["0", "4", "8", "16"])
def test_large_fft_tile_heights(monkeypatch, tile, fuse):
    """every tiling of the four-step passes (IRB_DEBUG lfft_tile: rows / columns per workgroup; 0 = one wave per item), with the
    real split as its own launch or fused into the tiled passes (lfft_fuse_split: bit 0 inverse, bit 1 forward),
    gives the same transform: 2^14 .. 2^21 points"""
    monkeypatch.setenv("IRB_DEBUG", f"lfft_tile={tile},lfft_fuse_split={fuse}")
    ib.lib().irb_release_cached_plans()
    for n in (1 << 14, 1 << 17, 1 << 20, 1 << 21):
        x = white_noise(n + int(tile), n)
        F = ib.fftTransform(x)
        ref = np.fft.rfft(x.astype(np.float64))
        got = _bins(F, n)
        assert np.sqrt(np.mean(np.abs(got - ref) ** 2) / np.mean(np.abs(ref) ** 2)) < 2e-6
        assert rel_rms(ib.fftInvTransform(F), x) < 1e-5
    ib.lib().irb_release_cached_plans()


def test_large_fft_plan_spectrum_buffer_contract():
    """irb_lfft_*: a forward transform INTO the plan's own spectrum buffer writes bins 0 .. N/2 only (the zeros the reference layout has
    above the Nyquist bin, fp/tools.hpp:84-87, were put there once).  A caller that stores above the Nyquist bin through the
    pointer irb_lfft_buffers hands out breaks that -- documented -- and irb_lfft_invalidate_spec re-arms the full write."""
    n = 1 << 14
    x = white_noise(77, n)
    plan = ib.LargeFft(n, 1)
    with ib.FdlEngine(1, 1, 16, 16) as io:                       # (only for its upload / download helpers: they take any device pointer)
        io.upload(plan.d_time, x)
        plan.forward()
        plan.synchronize()
        F = io.download(plan.d_spec, 2 * n)
        ref = np.fft.rfft(x.astype(np.float64))
        assert np.sqrt(np.mean(np.abs(_bins(F, n) - ref) ** 2) / np.mean(np.abs(ref) ** 2)) < 2e-6 and not F[n + 2:].any()
        io.upload(plan.d_spec + 4 * (n + 2), np.full(n - 2, 7.0, np.float32))      # the caller dirties floats N+2 .. 2N-1
        plan.forward()
        plan.synchronize()
        assert (io.download(plan.d_spec, 2 * n)[n + 2:] == 7.0).all()              # the documented contract: not rewritten by itself
        plan.invalidate_spec()
        for _ in range(2):                                                         # the next forward restores the zeros; the one after keeps them
            plan.forward()
            plan.synchronize()
            F = io.download(plan.d_spec, 2 * n)
            assert not F[n + 2:].any() and np.sqrt(np.mean(np.abs(_bins(F, n) - ref) ** 2) / np.mean(np.abs(ref) ** 2)) < 2e-6
    plan.close()


# ---- SURVEY section 8(f) ranks 3 and 4 ----------------------------------------------------------------
import os  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "convolve_cases.npz"))


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in GOLD.files}))
def test_convolve_nonperiodic_matches_golden_and_oracle(name):
    audio, ir = GOLD[name + "/audio"], GOLD[name + "/ir"]
    exp = GOLD[name + "/expected_f64"]
    y = ib.convolveNonPeriodic(audio, ir)
    ref, _ = oracle.convolve_nonperiodic(audio, ir)
    assert y.shape == exp.shape
    for c in range(y.shape[0]):
        assert rel_rms(y[c], exp[c]) < 1e-5 and rel_rms(y[c], ref[c]) < 1e-5


def test_convolve_nonperiodic_layout_and_empty():
    y = ib.convolveNonPeriodic(np.ones((3, 64), np.float32), np.ones((1, 16), np.float32))
    assert y.shape == (3, 79) and not y.any()
    assert not ib.convolveNonPeriodic(np.zeros((1, 0), np.float32), np.ones((1, 16), np.float32)).any()


def test_convolve_nonperiodic_long():
    x, h = white_noise(1, 700000)[None], decaying_ir(2, 300000)[None]
    import scipy.signal
    y = ib.convolveNonPeriodic(x, h)
    assert rel_rms(y[0], scipy.signal.fftconvolve(x[0].astype(np.float64), h[0].astype(np.float64))) < 1e-5


def test_convolve_nonperiodic_direct_and_staged_forms_agree(monkeypatch):
    """irb_convolve_nonperiodic stages pageable buffers through the plan's pinned memory (the default, and what serves the
    stereo-IR-to-mono layout); with IRB_OFFLINE_DIRECT=1 the copy engine moves the samples straight between the caller's registered
    buffers and the device.  Same kernels either way: identical bits, a poisoned result buffer fully overwritten, ragged lengths,
    audio and IR in one allocation."""
    from irbaboon_amd import _lib
    lib = _lib.lib()
    fpp = lambda arrs: (_lib.fp * len(arrs))(*[a.ctypes.data_as(_lib.fp) for a in arrs])
    for k, (Ca, Ci, La, Lh) in enumerate([(1, 1, 50000, 3001), (2, 2, 33001, 777), (2, 1, 12345, 999), (1, 2, 2000, 300), (1, 1, 5, 3)]):
        big = np.zeros(Ca * La + Ci * Lh + 64, np.float32)           # everything in one allocation, odd offsets
        xs = [big[3 + c * (La + 1):3 + c * (La + 1) + La] for c in range(Ca)]
        off = 3 + Ca * (La + 1) + 5
        hs = [big[off + c * (Lh + 1):off + c * (Lh + 1) + Lh] for c in range(Ci)]
        for c in range(Ca):
            xs[c][:] = white_noise(8600 + 10 * k + c, La)
        for c in range(Ci):
            hs[c][:] = decaying_ir(8650 + 10 * k + c, Lh)
        outs = []
        for direct in ("1", None):
            if direct:
                monkeypatch.setenv("IRB_OFFLINE_DIRECT", direct)
            else:
                monkeypatch.delenv("IRB_OFFLINE_DIRECT", raising=False)
            out = [np.full(La + Lh - 1, 1e30, np.float32) for _ in range(Ca)]
            _lib.check(lib.irb_convolve_nonperiodic(0, fpp(xs), Ca, La, fpp(hs), Ci, Lh, fpp(out)))
            outs.append(np.stack(out))
        assert np.array_equal(outs[0], outs[1]) and not (outs[0] == np.float32(1e30)).any()
        ref, _ = oracle.convolve_nonperiodic(np.stack(xs), np.stack(hs))
        for c in range(Ca):
            assert rel_rms(outs[0][c], ref[c]) < 1e-5
    lib.irb_release_cached_plans()


def test_ir_to_real_fft_raw_forms_agree_with_an_engine(monkeypatch):
    """irb_ir_to_real_fft_raw: one launch transforms the IR in the per-thread pinned staging (default) or where it lies in the caller's
    registered pages (IRB_OFFLINE_DIRECT=1), rows written straight to where the host reads them.  Both identical to the rows an FDL
    engine holds for the same IR (irb_fdl_get_ir_spectra), for partition sizes 16 ... 2048, ragged and exact lengths, growing and
    shrinking IRs."""
    for n, ps in [(700, 64), (96000, 256), (5, 16), (2048 * 3, 2048), (1000, 512), (96000, 512), (300, 256)]:
        h = decaying_ir(9000 + n, n)
        outs = []
        for direct in ("1", None):
            if direct:
                monkeypatch.setenv("IRB_OFFLINE_DIRECT", direct)
            else:
                monkeypatch.delenv("IRB_OFFLINE_DIRECT", raising=False)
            outs.append(ib.IRtoRealFFTRaw(h, ps))
        assert outs[0].shape == (n // ps + 1, 2 * ps) and np.array_equal(outs[0], outs[1])
        npart = (n + ps - 1) // ps
        with ib.FdlEngine(1, 1, ps, n) as eng:
            eng.set_ir(0, h[None])
            rows = eng.get_ir_spectra(0, 0, npart)
        assert np.array_equal(rows, outs[0][:npart])
    from irbaboon_amd import _lib
    _lib.lib().irb_release_cached_plans()


@pytest.mark.parametrize("n,ps", [(2048, 256), (2000, 256), (700, 64), (96000, 512)])
def test_ir_to_real_fft_raw_layout(n, ps):
    """ir::IRtoRealFFTRaw (fp/ir.cpp:106-147): len/ps + 1 rows of 2*ps floats, each the partition's FFT with
    im(0) := re(N/2); an exact multiple leaves a trailing all-zero row."""
    h = decaying_ir(n, n)
    raw = ib.IRtoRealFFTRaw(h, ps)
    assert raw.shape == (n // ps + 1, 2 * ps)
    N = 2 * ps
    for k in range((n + ps - 1) // ps):
        part = np.zeros(N)
        seg = h[k * ps:(k + 1) * ps]
        part[:seg.size] = seg
        F = np.fft.rfft(part)
        row = raw[k].astype(np.float64)
        got = row[0::2] + 1j * row[1::2]
        assert abs(got[0].real - F[0].real) < 1e-5 and abs(got[0].imag - F[ps].real) < 1e-5
        assert np.abs(got[1:] - F[1:ps]).max() < 1e-5
        # same numbers as the oracle's fftTransform of the partition, packed
        o = oracle.fft_transform(part.astype(np.float32))
        assert abs(row[1] - o[N]) < 1e-5 and np.abs(row[2:] - o[2:N]).max() < 1e-5
    if n % ps == 0:
        assert not raw[-1].any()


def test_ir_spectra_roundtrip_through_engine():
    """spectra exported by one engine and uploaded into another give the same convolution."""
    B, Lh, T = 128, 1000, 12
    irs = np.stack([decaying_ir(1, Lh), decaying_ir(2, Lh)])
    x = np.stack([white_noise(3, T * B), white_noise(4, T * B)])
    with ib.FdlEngine(1, 2, B, Lh) as a, ib.FdlEngine(1, 2, B, Lh) as b:
        a.set_ir(0, irs)
        ya = a.process(x)
        P = a.info.partitions
        sp = np.stack([a.get_ir_spectra(0, c, P) for c in range(2)])
        assert sp.shape == (2, P, 2 * B)
        b.set_ir_spectra(0, sp)
        assert np.array_equal(b.process(x), ya)
        raw = ib.IRtoRealFFTRaw(irs[0], B)
        assert np.allclose(raw[:P], sp[0], atol=1e-6)


def test_end_to_end_measurement_chain():
    """The plug-in's whole data path on the GPU, product APIs only: play the sweep through two 'rooms'
    (convolvePeriodic), extract both IRs (deconvolve, Source/PluginProcessor.cpp:306-312), derive the
    compensation filter (createIRFilt = deconvolve(target, base), :606-615) and run it through the
    processBlock-shaped convolver.  Checked against float64 maths in the band the sweep excites."""
    sr, Nsw = 48000, 65536
    sweep = exp_sine_sweep(49152, sr, 20.0, sr / 2, fade_from=(11.0 / 12.0) * sr / 2, n_total=Nsw)
    room_t, room_b = decaying_ir(1, 600), decaying_ir(2, 400)
    room_b[0] += 1.5                                     # base room well conditioned (it is divided by)
    cap_t = ib.convolvePeriodic(sweep[None], room_t[None], 256)[0, :Nsw]
    cap_b = ib.convolvePeriodic(sweep[None], room_b[None], 256)[0, :Nsw]
    ir_t = ib.deconvolve(cap_t, sweep, sr, smoothing=False)
    ir_b = ib.deconvolve(cap_b, sweep, sr, smoothing=False)
    band = slice(int(60 / sr * Nsw), int(18000 / sr * Nsw))

    def band_err(got, want_time):
        G = np.fft.rfft(got.astype(np.float64))
        W = np.fft.rfft(np.pad(want_time.astype(np.float64), (0, Nsw - want_time.size)))
        return np.sqrt(np.mean(np.abs(G[band] - W[band]) ** 2) / np.mean(np.abs(W[band]) ** 2))

    assert band_err(ir_t, room_t) < 2e-3 and band_err(ir_b, room_b) < 2e-3
    filt = ib.deconvolve(ir_t, ir_b, sr, smoothing=False)                  # createIRFilt without smoothing
    H = np.fft.rfft(np.pad(room_t.astype(np.float64), (0, Nsw - 600))) / np.fft.rfft(np.pad(room_b.astype(np.float64), (0, Nsw - 400)))
    F = np.fft.rfft(filt.astype(np.float64))
    assert np.sqrt(np.mean(np.abs(F[band] - H[band]) ** 2) / np.mean(np.abs(H[band]) ** 2)) < 5e-3
    # the real-time convolver uses the first 2048 samples of the filter (irPartitions fixed at 8, :225)
    rt = ib.RtConvolver(2, 256, 2048)
    rt.prepareToPlay(256)
    rt.setIR(filt)
    n = 256 * 40
    x = np.stack([white_noise(7, n), white_noise(8, n)])
    y = np.empty_like(x)
    for i in range(0, n, 256):
        blk = np.ascontiguousarray(x[:, i:i + 256])
        rt.processBlock(blk)
        y[:, i:i + 256] = blk
    want = np.stack([np.convolve(x[c].astype(np.float64), filt[:2048].astype(np.float64))[:n] for c in range(2)])
    s = 10 * 256
    assert rel_rms(y[:, 256 + s:], want[:, s:n - 256]) < 1e-5
