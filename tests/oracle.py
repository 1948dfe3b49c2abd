"""ctypes binding of the CPU oracle (oracle/libfp_oracle.so).  Test infrastructure:
imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(_ORACLE_DIR, "libfp_oracle.so")

_fpp = C.POINTER(C.POINTER(C.c_float))


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


def use_native():
    """bench.py's cpu_baseline: rebuild the oracle with -march=native ON this machine and load that build
    (call before the first lib()).  Returns the compile command line; falls back to the portable build
    (and says so) when there is no compiler."""
    global _SO, _lib
    try:
        subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR, "native"])
        flags = subprocess.check_output(["make", "-s", "-C", _ORACLE_DIR, "print-native-cflags"], text=True).strip()
        _SO, _lib = os.path.join(_ORACLE_DIR, "libfp_oracle_native.so"), None
        return flags
    except Exception:
        try:
            return subprocess.check_output(["make", "-s", "-C", _ORACLE_DIR, "print-cflags"], text=True).strip() + " (prebuilt)"
        except Exception:
            return "prebuilt libfp_oracle.so (-O3 -march=x86-64-v2)"


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    lib.fp_convolvePeriodic.restype = C.c_int
    lib.fp_convolvePeriodic.argtypes = [_fpp, C.c_int, C.c_int, _fpp, C.c_int, C.c_int, C.c_int, _fpp]
    lib.fp_convolveNonPeriodic.restype = C.c_int
    lib.fp_convolveNonPeriodic.argtypes = [_fpp, C.c_int, C.c_int, _fpp, C.c_int, C.c_int, _fpp]
    lib.fp_fft_create.restype = C.c_void_p
    lib.fp_fft_create.argtypes = [C.c_int]
    lib.fp_fft_destroy.argtypes = [C.c_void_p]
    lib.fp_fft_forward_real.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.fp_fft_inverse_real.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.fp_rt_create.restype = C.c_void_p
    lib.fp_rt_create.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.fp_rt_destroy.argtypes = [C.c_void_p]
    lib.fp_rt_prepareToPlay.argtypes = [C.c_void_p, C.c_int]
    lib.fp_rt_set_ir.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    lib.fp_rt_processBlock.argtypes = [C.c_void_p, _fpp, C.c_int]
    lib.fp_rt_processBlockBypassed.argtypes = [C.c_void_p, _fpp, C.c_int]
    lib.fp_rt_latency.restype = C.c_int
    lib.fp_rt_latency.argtypes = [C.c_void_p]
    lib.fp_rt_partitions.restype = C.c_int
    lib.fp_rt_partitions.argtypes = [C.c_void_p]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _planar(a):
    a = np.ascontiguousarray(np.atleast_2d(a), dtype=np.float32)
    ptrs = (C.POINTER(C.c_float) * a.shape[0])(*[a[i].ctypes.data_as(C.POINTER(C.c_float)) for i in range(a.shape[0])])
    return a, ptrs


def convolve_periodic(audio, ir, block=256):
    """fp::convolution::convolvePeriodic (fp/convolution.cpp:14-242) on planar
    float32 arrays [channels][samples]; returns [audio_ch][La+Lh-1]."""
    a, ap = _planar(audio)
    h, hp = _planar(ir)
    out = np.zeros((a.shape[0], max(a.shape[1] + h.shape[1] - 1, 0)), np.float32)
    _, op = _planar(out)
    # _planar copies only if needed; out is already contiguous float32, so op aliases it
    op = (C.POINTER(C.c_float) * out.shape[0])(*[out[i].ctypes.data_as(C.POINTER(C.c_float)) for i in range(out.shape[0])])
    rc = lib().fp_convolvePeriodic(ap, a.shape[0], a.shape[1], hp, h.shape[0], h.shape[1], block, op)
    return out, rc


def convolve_nonperiodic(audio, ir):
    a, ap = _planar(audio)
    h, hp = _planar(ir)
    out = np.zeros((a.shape[0], max(a.shape[1] + h.shape[1] - 1, 0)), np.float32)
    op = (C.POINTER(C.c_float) * out.shape[0])(*[out[i].ctypes.data_as(C.POINTER(C.c_float)) for i in range(out.shape[0])])
    rc = lib().fp_convolveNonPeriodic(ap, a.shape[0], a.shape[1], hp, h.shape[0], h.shape[1], op)
    return out, rc


def rfft(x, order):
    """performRealOnlyForwardTransform: returns complex bins 0..N/2."""
    n = 1 << order
    d = np.zeros(2 * n, np.float32)
    d[:n] = x
    f = lib().fp_fft_create(order)
    lib().fp_fft_forward_real(f, d.ctypes.data_as(C.POINTER(C.c_float)))
    lib().fp_fft_destroy(f)
    return d[0:n + 2:2] + 1j * d[1:n + 2:2]


def irfft(X, order):
    n = 1 << order
    d = np.zeros(2 * n, np.float32)
    d[0:n + 2:2] = X.real
    d[1:n + 2:2] = X.imag
    f = lib().fp_fft_create(order)
    lib().fp_fft_inverse_real(f, d.ctypes.data_as(C.POINTER(C.c_float)))
    lib().fp_fft_destroy(f)
    return d[:n].copy()


class RtProcessor:
    """IRBaboonAudioProcessor's convolution state machine
    (Source/PluginProcessor.cpp:57-82,164-234,408-562)."""

    def __init__(self, channels=2, block=256, ir_ring_len=2048):
        self.p = lib().fp_rt_create(channels, block, ir_ring_len)
        self.channels = channels

    def prepare_to_play(self, samples_per_block):
        lib().fp_rt_prepareToPlay(self.p, samples_per_block)

    def set_ir(self, ir):
        ir = np.ascontiguousarray(ir, np.float32)
        lib().fp_rt_set_ir(self.p, ir.ctypes.data_as(C.POINTER(C.c_float)), len(ir))

    def process_block(self, buf):
        """buf: float32 [channels][n], processed in place."""
        assert buf.dtype == np.float32 and buf.flags.c_contiguous and buf.shape[0] == self.channels
        ptrs = (C.POINTER(C.c_float) * buf.shape[0])(*[buf[i].ctypes.data_as(C.POINTER(C.c_float)) for i in range(buf.shape[0])])
        lib().fp_rt_processBlock(self.p, ptrs, buf.shape[1])

    def process_block_bypassed(self, buf):
        assert buf.dtype == np.float32 and buf.flags.c_contiguous and buf.shape[0] == self.channels
        ptrs = (C.POINTER(C.c_float) * buf.shape[0])(*[buf[i].ctypes.data_as(C.POINTER(C.c_float)) for i in range(buf.shape[0])])
        lib().fp_rt_processBlockBypassed(self.p, ptrs, buf.shape[1])

    @property
    def latency(self):
        return lib().fp_rt_latency(self.p)

    @property
    def partitions(self):
        return lib().fp_rt_partitions(self.p)

    def __del__(self):
        try:
            lib().fp_rt_destroy(self.p)
        except Exception:
            pass


# ---- config-5 chain (fp/convolution.cpp:351-546, fp/tools.cpp:321-369, fp/ir.cpp:13-18,85-103) ----
def _bind_cfg5():
    l = lib()
    if getattr(l, "_cfg5_bound", False):
        return l
    fp = C.POINTER(C.c_float)
    l.fp_nextPowerOfTwo.restype = C.c_int
    l.fp_nextPowerOfTwo.argtypes = [C.c_int]
    l.fp_fftTransform.restype = C.c_int
    l.fp_fftTransform.argtypes = [fp, C.c_int, fp]
    l.fp_fftTransformAmplPhase.restype = C.c_int
    l.fp_fftTransformAmplPhase.argtypes = [fp, C.c_int, fp]
    l.fp_fftInvTransform.argtypes = [fp, C.c_int, fp]
    l.fp_averagingFilter.argtypes = [fp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
    l.fp_deconvolve.restype = C.c_int
    l.fp_deconvolve.argtypes = [fp, C.c_int, fp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, fp]
    l.fp_shifteroo.argtypes = [fp, C.c_int]
    l.fp_invertFilter.restype = C.c_int
    l.fp_invertFilter.argtypes = [fp, C.c_int, C.c_int, fp]
    l._cfg5_bound = True
    return l


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def next_pow2(x):
    return _bind_cfg5().fp_nextPowerOfTwo(int(x))


def fft_transform(x, ampl_phase=False):
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(2 * next_pow2(x.size), np.float32)
    (_bind_cfg5().fp_fftTransformAmplPhase if ampl_phase else _bind_cfg5().fp_fftTransform)(_f(x), x.size, _f(out))
    return out


def fft_inv_transform(buf):
    buf = np.ascontiguousarray(buf, np.float32)
    out = np.zeros(buf.size // 2, np.float32)
    _bind_cfg5().fp_fftInvTransform(_f(buf), buf.size, _f(out))
    return out


def averaging_filter(buf, octave_fraction, sample_rate, log_avg=True, include_phase=True, include_amplitude=True):
    buf = np.ascontiguousarray(buf, np.float32).copy()
    _bind_cfg5().fp_averagingFilter(_f(buf), buf.size, octave_fraction, sample_rate, int(log_avg),
                                    int(include_phase), int(include_amplitude))
    return buf


def deconvolve(num, den, sample_rate, smoothing=True, include_phase=True, include_amplitude=True):
    num = np.ascontiguousarray(num, np.float32)
    den = np.ascontiguousarray(den, np.float32)
    out = np.zeros(next_pow2(max(num.size, den.size)), np.float32)
    _bind_cfg5().fp_deconvolve(_f(num), num.size, _f(den), den.size, float(sample_rate), int(smoothing),
                               int(include_phase), int(include_amplitude), _f(out))
    return out


def shifteroo(x):
    x = np.ascontiguousarray(x, np.float32).copy()
    _bind_cfg5().fp_shifteroo(_f(x), x.size)
    return x


def invert_filter(ir, sample_rate):
    ir = np.ascontiguousarray(ir, np.float32)
    out = np.zeros(next_pow2(ir.size), np.float32)
    _bind_cfg5().fp_invertFilter(_f(ir), ir.size, int(sample_rate), _f(out))
    return out


# ---- fp::ExpSineSweep (fp/ExpSineSweep.cpp) ----
def sweep_generate(duration, sr, lo, hi, gain_db=0.0, inverse=False):
    l = lib()
    dp = C.POINTER(C.c_double)
    for f in (l.fp_sweep_generate, l.fp_sweep_generateInv):
        f.restype = C.c_int
        f.argtypes = [C.c_double] * 5 + [dp]
    out = np.zeros(int(sr * duration) + 1, np.float64)
    n = (l.fp_sweep_generateInv if inverse else l.fp_sweep_generate)(duration, sr, lo, hi, gain_db, out.ctypes.data_as(dp))
    return out[:n].copy()


def sweep_index_at_freq(freq, duration, sr, lo, hi):
    l = lib()
    l.fp_sweep_getSampleIndexAtFreq.restype = C.c_int
    l.fp_sweep_getSampleIndexAtFreq.argtypes = [C.c_double] * 5
    return l.fp_sweep_getSampleIndexAtFreq(freq, duration, sr, lo, hi)


def sweep_freq_at_index(index, duration, sr, lo, hi):
    l = lib()
    l.fp_sweep_getFreqAtSampleIndex.restype = C.c_double
    l.fp_sweep_getFreqAtSampleIndex.argtypes = [C.c_int] + [C.c_double] * 4
    return l.fp_sweep_getFreqAtSampleIndex(index, duration, sr, lo, hi)


def sweep_lin_fadeout(sweep, index):
    l = lib()
    l.fp_sweep_linFadeout.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int]
    s = np.ascontiguousarray(sweep, np.float64).copy()
    l.fp_sweep_linFadeout(s.ctypes.data_as(C.POINTER(C.c_double)), s.size, index)
    return s
