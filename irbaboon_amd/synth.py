"""Bit-portable synthetic inputs for the convolver.

splitmix64 -> top 24 bits -> float32 in [-1, 1).  Pure integer arithmetic, so
the same seed gives the same samples on every machine (numpy's / libstdc++'s
float generators do not promise that).
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, n):
    """n successive outputs of splitmix64 started at `seed` (uint64 array)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def white_noise(seed, n):
    """Uniform white noise in [-1, 1), float32, 24-bit resolution."""
    u = (splitmix64(seed, n) >> np.uint64(40)).astype(np.int64)      # 24 bits
    return ((u - (1 << 23)).astype(np.float32) / np.float32(1 << 23)).astype(np.float32)


def decaying_ir(seed, n, decay=6.9):
    """h[i] = u_i * exp(-decay * i / n) (about -60 dB at the end), unit L2 norm."""
    u = white_noise(seed, n).astype(np.float64)
    h = u * np.exp(-decay * np.arange(n, dtype=np.float64) / max(n, 1))
    nrm = np.sqrt(np.sum(h * h))
    if nrm > 0:
        h = h / nrm
    return h.astype(np.float32)


def delta_ir(n=2048, offset=100):
    """The reference's default IR: generatePulse(2048, 100)
    (Source/PluginProcessor.cpp:46, fp/tools.cpp:235-241)."""
    h = np.zeros(n, np.float32)
    if offset >= n or offset < 0:
        offset = 0
    h[offset] = 1.0
    return h


def exp_sine_sweep(n_sweep, sample_rate, f_lo=20.0, f_hi=None, fade_from=None, n_total=None):
    """Farina exponential sine sweep as the plug-in builds it (input generator only):
    sweep[i] = sin(K*(exp(i/L)-1)), K = T*w1/ln(w2/w1), L = T/ln(w2/w1)
    (fp/ExpSineSweep.cpp:26-41,212-220), linear fade-out from frequency `fade_from`
    (:147-164, Source/PluginProcessor.cpp:29-30), followed by silence up to n_total samples."""
    f_hi = sample_rate / 2.0 if f_hi is None else f_hi
    T = float(n_sweep)
    w1, w2 = f_lo / sample_rate * 2 * np.pi, f_hi / sample_rate * 2 * np.pi
    K, L = T * w1 / np.log(w2 / w1), T / np.log(w2 / w1)
    i = np.arange(n_sweep, dtype=np.float64)
    s = np.sin(K * (np.exp(i / L) - 1.0))
    if fade_from is not None:
        wf = fade_from / sample_rate * 2 * np.pi
        idx = int(round(L * np.log(wf / w1)))              # getSampleIndexAtFreq (fp/ExpSineSweep.cpp:243-258)
        idx = min(max(idx, 0), n_sweep - 1)
        n = n_sweep - idx
        s[idx:] *= (n - np.arange(n)) / n
    out = np.zeros(n_total if n_total else n_sweep, np.float32)
    out[:n_sweep] = s.astype(np.float32)
    return out
