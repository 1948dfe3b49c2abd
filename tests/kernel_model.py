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


# ---- model of fft_wave.hpp (round 2): one wave per transform, IN PLACE (all lanes read, then all lanes write),
# ---- LDS twiddle table of 3N/4 entries (host half table + negated first quarter), compile-time unrolled stages with
# ---- j0 = 4j - 3k, split passes where lane b handles (b, M-b) for b < M/2 and the lane of b = 0 also the bin M/2

def half_table(N):
    return twiddle_table(N)[:N // 2]


def lds_table(ht, N):
    """what put_twiddle<N> leaves in LDS: W_N^j for j < 3N/4 (entries j + N/2 = -W_N^j for j < N/4)"""
    t = np.empty(3 * N // 4, np.complex64)
    t[:N // 2] = ht
    t[N // 2:] = -ht[:N // 4]
    return t


def wave_fft(buf, ht, N, inverse=False, lanes=64):
    M = len(buf)
    lt = lds_table(ht, N)
    buf = buf.astype(np.complex64).copy()
    Ns = 1
    while Ns < M:
        R = 4 if Ns * 4 <= M else 2
        nb = M // R
        tstep = N // (Ns * R)
        reads, tws = {}, {}
        for lane in range(lanes):                       # phase 1: every lane reads its points AND its twiddles
            for j in range(lane, nb, lanes):
                k = j & (Ns - 1)
                reads[j] = [buf[j + r * nb] for r in range(R)]
                if R == 4:
                    assert 3 * k * tstep < len(lt)      # the extended table covers every radix-4 index: no sign fix-up
                    tws[j] = [lt[k * tstep], lt[2 * k * tstep], lt[3 * k * tstep]]
                else:
                    tws[j] = [lt[2 * j]]                # closing radix-2 stage: W_N^(2j), j < M/2
        for j, v in reads.items():                      # phase 2: butterflies + in-place writes
            k = j & (Ns - 1)
            for r in range(1, R):
                w = tws[j][r - 1]
                if Ns > 1 or R == 2:
                    v[r] = np.complex64(v[r] * (np.conj(w) if inverse else w))
            if R == 4:
                s0, d0, s1, e = v[0] + v[2], v[0] - v[2], v[1] + v[3], v[1] - v[3]
                d1 = e * (1j if inverse else -1j)
                o = [s0 + s1, d0 + d1, s0 - s1, d0 - d1]
                j0 = 4 * j - 3 * k
            else:
                o = [v[0] + v[1], v[0] - v[1]]
                j0 = j
                assert Ns == M // 2
            for r in range(R):
                buf[j0 + r * Ns] = np.complex64(o[r])
        Ns *= R
    return buf


def wave_rfft_packed(x, N, ht):
    M = N // 2
    lt = lds_table(ht, N)
    Z = wave_fft((x[0::2] + 1j * x[1::2]).astype(np.complex64), ht, N)
    X = np.empty(M, np.complex64)
    for b in range(M // 2):                             # b = lane + 64 i < M/2; all reads happen before the stores
        p, q = Z[b], Z[(M - b) & (M - 1)]
        if b == 0:
            X[0] = np.complex64(complex(p.real + p.imag, p.real - p.imag))
            X[M // 2] = np.conj(Z[M // 2])              # the self-paired bin, emitted by the lane of b = 0
        else:
            e = np.complex64(complex(0.5 * (p.real + q.real), 0.5 * (p.imag - q.imag)))
            o = np.complex64(complex(0.5 * (p.imag + q.imag), -0.5 * (p.real - q.real)))
            t = np.complex64(lt[b] * o)
            X[b] = np.complex64(e + t)
            X[M - b] = np.complex64(complex(e.real - t.real, t.imag - e.imag))
    return X


def wave_irfft_packed(Y, N, ht):
    M = N // 2
    lt = lds_table(ht, N)
    Z = Y.astype(np.complex64).copy()
    for b in range(M // 2):
        p, q = Y[b], Y[(M - b) & (M - 1)]
        if b == 0:
            Z[0] = np.complex64(complex(p.real + p.imag, p.real - p.imag))
            Z[M // 2] = np.complex64(2 * np.conj(Y[M // 2]))
        else:
            e = complex(p.real + q.real, p.imag - q.imag)
            d = np.complex64(complex(p.real - q.real, p.imag + q.imag))
            o = np.complex64(d * np.conj(lt[b]))
            Z[b] = np.complex64(complex(e.real - o.imag, e.imag + o.real))
            Z[M - b] = np.complex64(complex(e.real + o.imag, o.real - e.imag))
    z = wave_fft(Z, ht, N, inverse=True) * np.float32(1.0 / N)
    y = np.empty(N, np.float32)
    y[0::2] = z.real
    y[1::2] = z.imag
    return y


# ---- model of k_lfft_rows_split's row pairing (lfft_kernels.hpp): which rows a workgroup owns, and where the mirror
# ---- M - e of every bin e = k1 + M1*k2 sits in its tile

def rows_split_layout(M1, M2, R):
    """yields (group, slot r, row k1, k2, mirror slot, mirror column) for every bin, as the kernel computes them"""
    H = R // 2
    for g in range(M1 // R):
        P0 = g * H

        def row_of(r):
            p = P0 + r if r < H else P0 + (R - 1 - r)
            return p if r < H else (M1 // 2 if p == 0 else M1 - p)
        for r in range(R):
            for k2 in range(M2):
                self_row = P0 == 0 and r in (0, R - 1)
                mr = r if self_row else R - 1 - r
                mc = ((M2 - k2) & (M2 - 1)) if (self_row and r == 0) else M2 - 1 - k2
                yield g, r, row_of(r), k2, row_of(mr), mc


# ---- csrc/fdl_mfma.hpp: the time axis of one bin as Toeplitz-block products ---------------------------------------
MFMA_SEG, MFMA_JC = 16, 9
MFMA_NC, MFMA_NH = MFMA_JC + MFMA_SEG - 1, MFMA_JC + 2


def mfma_chunks(np_):
    """chunks j = 0 .. J-1 of 16 partitions whose Toeplitz blocks M_j[k][i] = H[16 j + i - k] cover every m in [0, np)"""
    return (np_ + MFMA_SEG - 2) // MFMA_SEG + 1 if np_ > 0 else 0


def mfma_time_axis(X, H, T, splits=1):
    """k_cmac_mfma for ONE bin: X[tau] for tau in (-np, T) (dict-like: callable tau -> complex, zero outside), H[m], m < np.
    Walks tiles of 256 output blocks, partition splits and stages of MFMA_JC chunks exactly as the kernel does -- stage windows
    (tau_lo, m_lo), the LDS images' index formulas, the operand positions of lane (n, kq) at (chunk j, k group g), the output
    register map -- and returns [splits][T] partial sums (float64: this checks indices, not rounding)."""
    np_ = len(H)
    J = mfma_chunks(np_)
    out = np.zeros((splits, T), np.complex128)
    x_at = lambda tau: X(tau) if -np_ < tau < T else 0.0
    h_at = lambda m: H[m] if 0 <= m < np_ else 0.0
    for tile in range((T + 255) // 256):
        s0 = tile * MFMA_SEG
        for split in range(splits):
            jb, je = J * split // splits, J * (split + 1) // splits
            acc = np.zeros((MFMA_SEG, MFMA_SEG), np.complex128)          # O[n][i]
            for ja in range(jb, je, MFMA_JC):
                jc = min(MFMA_JC, je - ja)
                tau_lo, m_lo = MFMA_SEG * (s0 - ja - jc + 1), MFMA_SEG * ja - MFMA_SEG
                # the stage images: X as [k][X row], H as [m - m_lo]; staging thread srow -> k = srow % 16, row 4 it + srow / 16
                lx = np.zeros(MFMA_SEG * MFMA_NC, np.complex128)
                lh = np.zeros(MFMA_SEG * MFMA_NH, np.complex128)
                for srow in range(64):
                    sk, sc = srow % MFMA_SEG, srow // MFMA_SEG
                    for it in range((MFMA_NC + 3) // 4):
                        c = 4 * it + sc
                        if c < MFMA_NC:
                            ok = c < jc + MFMA_SEG - 1
                            lx[sk * MFMA_NC + sc + 4 * it] = x_at(tau_lo + srow + 64 * it) if ok else 0.0
                    for it in range((MFMA_NH + 3) // 4):
                        if 4 * it + sc < MFMA_NH:
                            ok = 4 * it + sc < jc + 2
                            lh[srow + 64 * it] = h_at(m_lo + srow + 64 * it) if ok else 0.0
                # the multiply: lane (n, kq), chunk j, k group g
                for j in range(jc):
                    for g in range(4):
                        for kq in range(4):
                            xs = np.array([lx[kq * MFMA_NC + n + jc - 1 + 4 * g * MFMA_NC - j] for n in range(MFMA_SEG)])
                            hs = np.array([lh[MFMA_SEG + i - kq + MFMA_SEG * j - 4 * g] for i in range(MFMA_SEG)])
                            acc += np.outer(xs, hs)                       # one k of v_mfma_f32_16x16x4: O[n][i] += X(n, k) * H(k, i)
            for n in range(MFMA_SEG):
                for i in range(MFMA_SEG):
                    t = 256 * tile + MFMA_SEG * n + i
                    if t < T:
                        out[split, t] = acc[n, i]
    return out
