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
MFMA_SEG, MFMA_TILE = 16, 256


def mfma_chunks(np_):
    """chunks j = 0 .. J-1 of 16 partitions whose Toeplitz blocks M_j[k][i] = H[16 j + i - k] cover every m in [0, np)"""
    return (np_ + MFMA_SEG - 2) // MFMA_SEG + 1 if np_ > 0 else 0


def mfma_hist(P):
    return MFMA_SEG * (mfma_chunks(P) + 3)


def mfma_taps(P):
    return MFMA_SEG * (mfma_chunks(P) + 4)


def mfma_plan(T, units, P, wgs=256, min_chunks=6, res=1):
    """fdl_engine.hip: mfma_plan -- the launches (tile_first, tile_count, splits) of one call: whole rounds of workgroups (res per CU,
    each 1 / res of the partitions), then the remaining tiles with their partitions split over all resident slots.
    units = streams x bin groups = workgroups per tile at one split"""
    by_work = max(1, mfma_chunks(P) // min_chunks)
    splits_for = lambda w: max(1, min(by_work, wgs * res // max(1, w)))
    per_round = wgs // units
    out = []

    def run(first, ntiles):
        if ntiles <= 0:
            return
        full = ntiles // per_round * per_round if per_round > 0 else ntiles
        if full == 0:
            out.append((first, ntiles, splits_for(ntiles * units) if ntiles * units <= wgs else 1))
            return
        out.append((first, full, min(by_work, res)))
        if ntiles > full:                                            # the remainder's workgroups are few and short: down to three chunks per split
            out.append((first + full, ntiles - full, max(1, min(max(1, mfma_chunks(P) // 3), wgs * res // max(1, (ntiles - full) * units)))))

    run(0, (T + MFMA_TILE - 1) // MFMA_TILE)
    return out


def mfma_xt_float(t, c):
    """float index of component c (0 re, 1 im) of block t in Xt: inside every segment of 16 blocks (32 floats) block 4 g + kq sits
    at 8 kq + 4 c + g"""
    return ((t & ~15) << 1) | ((t & 3) << 3) | (c << 2) | ((t >> 2) & 3)


def mfma_xt_slot_block(p):
    """the 8-byte slot p < 16 of a segment holds component (p >> 1) & 1 of this block of the segment and of that + 4"""
    return ((p & 1) << 3) | (p >> 2)


def mfma_layouts(X, H, T, P):
    """The operands as the kernels that produce them lay them out for ONE bin: Xt = [history | the call's blocks | zeros up to whole
    tiles] as FLOATS, every segment of 16 in mfma_xt_float order, written slot by slot the way k_push_fft_batch_t does (8-byte slot p
    = one component of two blocks); Ht = [16 zeros | taps | zeros] (k_rows_to_columns).  X: callable tau -> complex for tau in
    [-hist, T); H: the np taps in use (np <= P)."""
    hist, taps = mfma_hist(P), mfma_taps(P)
    Tpad = (T + MFMA_TILE - 1) // MFMA_TILE * MFMA_TILE
    lin = np.zeros(hist + Tpad, np.complex128)
    for tau in range(-hist, T):
        lin[hist + tau] = X(tau)
    lin[hist + T:] = 12345.0 + 6789.0j                 # what an earlier, longer call left behind: finite, must not matter
    xt = np.full(2 * len(lin), np.nan)
    for seg in range(0, len(lin), 16):                 # (hist is a multiple of 16: segments of the call and of the buffer coincide)
        for p in range(16):
            c, t = (p >> 1) & 1, mfma_xt_slot_block(p)
            for k, blk in enumerate((t, t + 4)):
                v = lin[seg + blk]
                xt[2 * (seg + p) + k] = v.imag if c else v.real
    for e in range(len(lin)):                          # the slot writer and the per-float formula (partial segments, N >= 1024) agree
        assert xt[mfma_xt_float(e, 0)] == lin[e].real and xt[mfma_xt_float(e, 1)] == lin[e].imag
    ht = np.zeros(taps, np.complex128)
    ht[16:16 + len(H)] = H
    return xt, ht, hist


def mfma_wave(xt, ht, hist, tile, jb, je):
    """mfma_chunk_run for one wave (one bin of one tile, chunks [jb, je)): byte offsets of the buffer loads (per-lane voffset, scalar
    offset, immediate), which lanes load (the taps: lanes 0 .. 30, one 8-byte load; a chunk's new segment of blocks: the four lanes
    n = 0, two 16-byte loads: four real parts, four imaginary parts), ds_bpermute handing the taps to the lanes that multiply with
    them, the DPP row shift that moves the X window along, the three register sets that rotate, the three real products of a
    complex one, the MFMA's operand and result maps.  Returns O[n][i] (16 x 16)."""
    lanes = [(l % 16, l // 16) for l in range(64)]                 # (n or i, kq)
    x_elem0 = hist + MFMA_TILE * tile
    sx = lambda j: (x_elem0 - 16 * j) * 8
    sh = lambda j: 16 * j * 8

    def ld(buf, byte):
        assert byte % 8 == 0 and 0 <= byte // 8 < len(buf)          # nothing the kernel addresses is out of range
        return buf[byte // 8]

    def ld32(byte):                                                 # two buffer_load_dwordx4: re of the lane's four k groups, then im
        assert byte % 16 == 0 and 0 <= byte // 4 and byte // 4 + 8 <= len(xt)
        f = xt[byte // 4: byte // 4 + 8]
        return list(f[:4]), list(f[4:])

    class Set:
        def __init__(self):
            self.xr = [[0.0] * 64 for _ in range(4)]
            self.xi = [[0.0] * 64 for _ in range(4)]
            self.h = [[0.0] * 64 for _ in range(4)]
            self.taps = [0.0] * 64

    def load_taps(s, j):                                            # lanes 0 .. 30: tap 16 j - 15 + l at Ht element 16 j + 1 + l; the others get zeros
        for l in range(64):
            s.taps[l] = ld(ht, (l + 1) * 8 + sh(j)) if l < 31 else 0.0

    def load_new(s, j):                                             # lanes n = 0: their four k groups of the segment chunk j adds; the others get zeros
        for l, (n, kq) in enumerate(lanes):
            re, im = ld32(4 * kq * 8 + sx(j)) if n == 0 else ([0.0] * 4, [0.0] * 4)
            for g in range(4):
                s.xr[g][l], s.xi[g][l] = re[g], im[g]

    def spread_taps(s):                                             # ds_bpermute: lane (i, kq) takes the tap lane 15 + i - kq - 4 g holds
        for g in range(4):
            for l, (i, kq) in enumerate(lanes):
                src = 15 + i - kq - 4 * g
                assert 0 <= src <= 30
                s.h[g][l] = s.taps[src]

    def shift(nxt, cur):                                            # v_mov_b32_dpp row_shr:1, bound_ctrl off: lanes n = 0 keep what they loaded
        for g in range(4):
            nxt.xr[g] = [nxt.xr[g][l] if n == 0 else cur.xr[g][l - 1] for l, (n, kq) in enumerate(lanes)]
            nxt.xi[g] = [nxt.xi[g][l] if n == 0 else cur.xi[g][l - 1] for l, (n, kq) in enumerate(lanes)]

    acc_r, acc_n, acc_s = np.zeros((16, 16)), np.zeros((16, 16)), np.zeros((16, 16))

    def multiply(s):
        for g in range(4):
            for l, (n, kq) in enumerate(lanes):                     # A operand lane (row n, k = kq), B operand lane (k = kq, column i)
                for l2, (i, kq2) in enumerate(lanes):
                    if kq2 == kq:
                        hr, hi = s.h[g][l2].real, s.h[g][l2].imag
                        acc_r[n, i] += s.xr[g][l] * hr
                        acc_n[n, i] += s.xi[g][l] * hi
                        acc_s[n, i] += (s.xr[g][l] + s.xi[g][l]) * (hr + hi)

    def step(j, cur, nxt):
        spread_taps(nxt)
        shift(nxt, cur)
        multiply(cur)
        load_new(cur, j + 3)
        load_taps(cur, j + 3)

    A, B, C = Set(), Set(), Set()
    for l, (n, kq) in enumerate(lanes):                             # the whole window of the first chunk: every lane its own 32 bytes
        re, im = ld32((16 * n + 4 * kq) * 8 + sx(jb))
        for g in range(4):
            A.xr[g][l], A.xi[g][l] = re[g], im[g]
    load_taps(A, jb)
    load_new(B, jb + 1)
    load_taps(B, jb + 1)
    load_new(C, jb + 2)
    load_taps(C, jb + 2)
    spread_taps(A)
    j = jb
    while j + 3 <= je:
        step(j, A, B)
        step(j + 1, B, C)
        step(j + 2, C, A)
        j += 3
    if j < je:
        step(j, A, B)
        if j + 1 < je:
            multiply(B)
    return (acc_r - acc_n) + 1j * (acc_s - acc_r - acc_n)


def mfma_time_axis(X, H, T, P=None, units=16, wgs=256):
    """k_cmac_mfma + the launch plan with its partial-spectrum regions + k_ifft_ola_run's region look-up and sum over partial
    spectra, for ONE bin: returns Y[t], t < T (float64: this checks indices, not rounding)."""
    np_ = len(H)
    P = np_ if P is None else P
    xt, ht, hist = mfma_layouts(X, H, T, P)
    J = mfma_chunks(np_)
    plan = mfma_plan(T, units, P, wgs)
    assert 1 <= len(plan) <= 2
    t0 = [l[0] * MFMA_TILE for l in plan] + [T] * (3 - len(plan))               # reg_t0
    parts = [np.zeros((t0[r + 1] - t0[r], plan[r][2]), np.complex128) for r in range(len(plan))]
    for r, (tile_first, tile_count, splits) in enumerate(plan):
        for tile in range(tile_first, tile_first + tile_count):
            for split in range(splits):
                jb, je = J * split // splits, J * (split + 1) // splits
                O = mfma_wave(xt, ht, hist, tile, jb, je) if jb < je else np.zeros((16, 16), np.complex128)
                for n in range(16):                                 # accumulator register v of lane (i, kq): O[4 kq + v][i]
                    for i in range(16):
                        t = MFMA_TILE * tile + 16 * n + i
                        if t < T:
                            parts[r][t - t0[r], split] = O[n, i]
    out = np.zeros(T, np.complex128)
    for t in range(T):
        r = 1 if t >= t0[1] else 0                                  # k_ifft_ola_run's look-up
        out[t] = parts[r][t - t0[r]].sum()
    return out


def xcd_contiguous(bid, nwg):
    """fdl_kernels.hpp: xcd_contiguous_id -- workgroup bid runs on XCD bid % 8; the ids of one XCD become a contiguous range"""
    q, r, xcd, local = nwg >> 3, nwg & 7, bid & 7, bid >> 3
    return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + local


def mfma_one_grid_decode(block_idx, grid, plan, S, groups):
    """k_cmac_mfma's first lines when ONE grid carries both launches of a plan: workgroups [0, wgs0) are the first region, the rest
    (last in dispatch order) the second; each region decodes (bin group, split, tile, stream) from its own XCD-contiguous id.
    plan = [(tile_first, tile_count, splits), ...] (one or two entries).  Returns (region, stream, tile, split, bin group)."""
    wgs0 = S * plan[0][1] * plan[0][2] * groups
    second = len(plan) == 2 and block_idx >= wgs0
    tile_first, tile_count, splits = plan[1] if second else plan[0]
    wg = xcd_contiguous(block_idx - wgs0, grid - wgs0) if second else xcd_contiguous(block_idx, wgs0 if len(plan) == 2 else grid)
    return (1 if second else 0, wg // (groups * splits * tile_count), tile_first + (wg // (groups * splits)) % tile_count,
            (wg // groups) % splits, wg % groups)

