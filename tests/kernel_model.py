"""Numpy model of the index/twiddle arithmetic used by the HIP kernels in
irbaboon_amd/csrc/fdl_kernels.hip (Stockham radix-4/2 stages, the real-FFT
split passes and the DC/Nyquist packing).  It exists so that the kernels'
index math can be checked on a machine without a GPU (tests/test_kernel_model.py);
it is test infrastructure, never imported by the product path.
"""
import numpy as np


def twiddle_table(N):
    """W_N^j for j in [0, N): computed in float64, stored as complex64 (what the
    engine uploads to the device)."""
    j = np.arange(N, dtype=np.float64)
    return np.exp(-2j * np.pi * j / N).astype(np.complex64)


def stockham(z, tw, N, inverse=False):
    """M-point complex FFT (M = len(z), N = 2M is the twiddle table length) as the
    kernel does it: radix-4 stages while 4*Ns <= M, then one radix-2 stage.
    One 'thread' j per butterfly; ping-pong buffers; autosort (no bit reversal)."""
    M = len(z)
    a = z.astype(np.complex64).copy()
    b = np.empty_like(a)
    Ns = 1
    while Ns < M:
        R = 4 if Ns * 4 <= M else 2
        stride = M // R
        tstep = N // (Ns * R)          # W_{Ns*R}^k == W_N^{k*tstep}
        for j in range(stride):        # thread j
            k = j % Ns
            v = [a[j + r * stride] for r in range(R)]
            for r in range(1, R):
                w = tw[(k * r * tstep) % N]
                if inverse:
                    w = np.conj(w)
                v[r] = np.complex64(v[r] * w)
            if R == 4:
                s0 = v[0] + v[2]; d0 = v[0] - v[2]
                s1 = v[1] + v[3]; d1 = v[1] - v[3]
                d1 = d1 * (1j if inverse else -1j)
                o = [s0 + s1, d0 + d1, s0 - s1, d0 - d1]
            else:
                o = [v[0] + v[1], v[0] - v[1]]
            j0 = (j // Ns) * Ns * R + k
            for r in range(R):
                b[j0 + r * Ns] = np.complex64(o[r])
        a, b = b, a
        Ns *= R
    return a


def rfft_packed(x, N, tw):
    """Forward: N real samples -> M=N/2 packed complex bins,
    bin 0 = (DC, Nyquist) (packing precedent: reference fp/ir.cpp:123-124)."""
    M = N // 2
    z = (x[0::2] + 1j * x[1::2]).astype(np.complex64)
    Z = stockham(z, tw, N)
    X = np.empty(M, np.complex64)
    for b in range(M):                 # thread b
        if b == 0:
            X[0] = np.complex64(complex(Z[0].real + Z[0].imag, Z[0].real - Z[0].imag))
        else:
            zc = np.conj(Z[M - b])
            e = (Z[b] + zc) * np.float32(0.5)
            o = (Z[b] - zc) * np.complex64(-0.5j)
            X[b] = np.complex64(e + tw[b] * o)
    return X


def irfft_packed(Y, N, tw):
    """Inverse of rfft_packed including the 1/N scale the reference relies on
    (fp/tools.cpp:366, fp/convolution.cpp:332-333)."""
    M = N // 2
    Z = np.empty(M, np.complex64)
    for b in range(M):
        if b == 0:
            Z[0] = np.complex64(complex(Y[0].real + Y[0].imag, Y[0].real - Y[0].imag))
        else:
            yc = np.conj(Y[M - b])
            e = Y[b] + yc
            o = (Y[b] - yc) * np.conj(tw[b])
            Z[b] = np.complex64(e + 1j * o)
    z = stockham(Z, tw, N, inverse=True) * np.float32(1.0 / N)
    y = np.empty(N, np.float32)
    y[0::2] = z.real
    y[1::2] = z.imag
    return y


def cmul_packed(a, c):
    """Bin-wise product of two packed spectra: bin 0 holds two real bins."""
    out = (a * c).astype(np.complex64)
    out[0] = np.complex64(complex(a[0].real * c[0].real, a[0].imag * c[0].imag))
    return out


# ---- model of fft_wave.hpp: one wave per transform, IN PLACE (all lanes read, then all lanes write),
# ---- half twiddle table with sign symmetry, split passes that handle bins b and M-b in one lane

def half_table(N):
    return twiddle_table(N)[:N // 2]


def tw_half(ht, N, i):
    w = ht[i & (N // 2 - 1)]
    return -w if (i & (N // 2)) else w


def wave_fft(buf, ht, N, inverse=False, lanes=64):
    M = len(buf)
    buf = buf.astype(np.complex64).copy()
    Ns = 1
    while Ns < M:
        R = 4 if Ns * 4 <= M else 2
        nb = M // R
        tstep = N // (Ns * R)
        reads = {}
        for lane in range(lanes):                       # phase 1: every lane reads all its points
            for j in range(lane, nb, lanes):
                reads[j] = [buf[j + r * nb] for r in range(R)]
        for j, v in reads.items():                      # phase 2: butterflies + in-place writes
            k = j & (Ns - 1)
            for r in range(1, R):
                w = tw_half(ht, N, k * r * tstep)
                v[r] = np.complex64(v[r] * (np.conj(w) if inverse else w))
            if R == 4:
                s0, d0, s1, e = v[0] + v[2], v[0] - v[2], v[1] + v[3], v[1] - v[3]
                d1 = e * (1j if inverse else -1j)
                o = [s0 + s1, d0 + d1, s0 - s1, d0 - d1]
            else:
                o = [v[0] + v[1], v[0] - v[1]]
            j0 = (j - k) * R + k
            for r in range(R):
                buf[j0 + r * Ns] = np.complex64(o[r])
        Ns *= R
    return buf


def wave_rfft_packed(x, N, ht):
    M = N // 2
    Z = wave_fft((x[0::2] + 1j * x[1::2]).astype(np.complex64), ht, N)
    X = np.empty(M, np.complex64)
    for b in range(M // 2 + 1):
        if b == 0:
            X[0] = np.complex64(complex(Z[0].real + Z[0].imag, Z[0].real - Z[0].imag))
        else:
            p, q = Z[b], Z[M - b]
            e = np.complex64(complex(0.5 * (p.real + q.real), 0.5 * (p.imag - q.imag)))
            o = np.complex64(complex(0.5 * (p.imag + q.imag), -0.5 * (p.real - q.real)))
            t = np.complex64(tw_half(ht, N, b) * o)
            X[b] = np.complex64(e + t)
            if b != M - b:
                X[M - b] = np.complex64(complex(e.real - t.real, t.imag - e.imag))
    return X


def wave_irfft_packed(Y, N, ht):
    M = N // 2
    Z = Y.astype(np.complex64).copy()
    for b in range(M // 2 + 1):
        if b == 0:
            Z[0] = np.complex64(complex(Y[0].real + Y[0].imag, Y[0].real - Y[0].imag))
        else:
            p, q = Y[b], Y[M - b]
            e = complex(p.real + q.real, p.imag - q.imag)
            d = np.complex64(complex(p.real - q.real, p.imag + q.imag))
            o = np.complex64(d * np.conj(tw_half(ht, N, b)))
            Z[b] = np.complex64(complex(e.real - o.imag, e.imag + o.real))
            if b != M - b:
                Z[M - b] = np.complex64(complex(e.real + o.imag, o.real - e.imag))
    z = wave_fft(Z, ht, N, inverse=True) * np.float32(1.0 / N)
    y = np.empty(N, np.float32)
    y[0::2] = z.real
    y[1::2] = z.imag
    return y
