"""The numpy model of the HIP kernels' index/twiddle arithmetic (tests/kernel_model.py)
against numpy's FFT: catches Stockham / split-pass / packing mistakes without a GPU."""
import numpy as np
import pytest

from kernel_model import cmul_packed, irfft_packed, rfft_packed, stockham, twiddle_table


@pytest.mark.parametrize("N", [32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_model(N):
    rng = np.random.default_rng(N)
    tw, M = twiddle_table(N), N // 2
    z = (rng.standard_normal(M) + 1j * rng.standard_normal(M)).astype(np.complex64)
    tol = 4e-7 * N
    assert np.abs(stockham(z, tw, N) - np.fft.fft(z.astype(np.complex128))).max() < tol
    assert np.abs(stockham(z, tw, N, True) - np.fft.ifft(z.astype(np.complex128)) * M).max() < tol
    x = rng.standard_normal(N).astype(np.float32)
    X, R = rfft_packed(x, N, tw), np.fft.rfft(x.astype(np.float64))
    assert np.abs(X[1:] - R[1:M]).max() < tol
    assert abs(X[0].real - R[0].real) < tol and abs(X[0].imag - R[M].real) < tol
    assert np.abs(irfft_packed(X, N, tw) - x).max() < 2e-6
    h = rng.standard_normal(N).astype(np.float32)
    y = irfft_packed(cmul_packed(X, rfft_packed(h, N, tw)), N, tw)
    ref = np.fft.irfft(np.fft.rfft(x.astype(np.float64)) * np.fft.rfft(h.astype(np.float64)))
    assert np.abs(y - ref).max() / np.abs(ref).max() < 2e-6


@pytest.mark.parametrize("N", [32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_wave_model(N):
    """fft_wave.hpp: in-place one-wave transform, 3N/4-entry LDS twiddle table, paired split passes."""
    from kernel_model import half_table, wave_fft, wave_irfft_packed, wave_rfft_packed
    rng = np.random.default_rng(N + 1)
    ht, M = half_table(N), N // 2
    z = (rng.standard_normal(M) + 1j * rng.standard_normal(M)).astype(np.complex64)
    tol = 4e-7 * N
    assert np.abs(wave_fft(z, ht, N) - np.fft.fft(z.astype(np.complex128))).max() < tol
    assert np.abs(wave_fft(z, ht, N, True) - np.fft.ifft(z.astype(np.complex128)) * M).max() < tol
    x = rng.standard_normal(N).astype(np.float32)
    X, R = wave_rfft_packed(x, N, ht), np.fft.rfft(x.astype(np.float64))
    assert np.abs(X[1:] - R[1:M]).max() < tol
    assert abs(X[0].real - R[0].real) < tol and abs(X[0].imag - R[M].real) < tol
    assert np.abs(wave_irfft_packed(X, N, ht) - x).max() < 2e-6


@pytest.mark.parametrize("M1,M2,R", [(64, 64, 4), (64, 128, 8), (512, 1024, 4), (256, 256, 8)])
def test_fused_split_row_pairing(M1, M2, R):
    """k_lfft_rows_split: every bin e = k1 + M1*k2 of Z is owned by exactly one workgroup, and the tile position the
    kernel reads as its mirror really holds Z[M - e] (e = 0 pairs with itself: DC / Nyquist)."""
    from kernel_model import rows_split_layout
    M = M1 * M2
    seen = np.zeros(M, np.int64)
    for g, r, k1, k2, mk1, mk2 in rows_split_layout(M1, M2, R):
        e = k1 + M1 * k2
        seen[e] += 1
        assert mk1 + M1 * mk2 == (M - e) % M, (g, r, k1, k2)
    assert np.all(seen == 1)


@pytest.mark.parametrize("np_,P,T,units,wgs", [(1, 1, 5, 16, 256), (15, 15, 40, 16, 256), (16, 16, 256, 16, 256), (17, 40, 300, 16, 256),
                                               (150, 150, 257, 16, 256), (375, 375, 70, 16, 256), (160, 375, 33, 4, 256),
                                               (100, 100, 600, 16, 32), (33, 33, 1100, 8, 16)])
def test_toeplitz_block_time_axis(np_, P, T, units, wgs):
    """fdl_mfma.hpp: one bin's time axis as sums of 16 x 16 Toeplitz-block products -- the transposed layouts (history, zero taps in front
    and behind, stale finite values beyond the call), tiles, partition splits, the two-launch plan with its two partial-spectrum
    regions, the buffer loads' byte offsets, the register sets that swap roles and the MFMA operand /
    result maps, restated in numpy -- equals the direct sum Y[t] = sum_m H[m] X[t - m] over the FDL history and the call's blocks.
    (wgs < 256 stands in for a smaller chip so that small cases reach the two-launch branch.)"""
    from kernel_model import mfma_chunks, mfma_hist, mfma_plan, mfma_time_axis
    rng = np.random.default_rng(1000 * np_ + T)
    H = rng.standard_normal(np_) + 1j * rng.standard_normal(np_)
    hist = mfma_hist(P)
    xs = rng.standard_normal(hist + T) + 1j * rng.standard_normal(hist + T)          # X[tau] at index tau + hist: history first
    X = lambda tau: xs[tau + hist]
    got = mfma_time_axis(X, H, T, P, units, wgs)
    ref = np.array([sum(H[m] * X(t - m) for m in range(np_)) for t in range(T)])
    assert np.abs(got - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    J = mfma_chunks(np_)
    assert 16 * (J - 1) - 15 <= np_ - 1 < 16 * J            # the last chunk is needed, one more would multiply zeros only
    assert hist >= np_ - 1                                   # the history columns hold every block an output can depend on
    plan = mfma_plan(T, units, P, wgs)
    assert sum(l[1] for l in plan) == (T + 255) // 256 and all(a[0] + a[1] == b[0] for a, b in zip(plan, plan[1:]))     # the launches tile the call


@pytest.mark.parametrize("T,S,groups,P,wgs", [(4125, 1, 16, 375, 256), (1280, 1, 2, 100, 8), (5000, 2, 8, 3750, 256), (3750, 1, 16, 375, 256), (700, 3, 4, 60, 64)])
def test_one_grid_carries_both_regions(T, S, groups, P, wgs):
    """fdl_mfma.hpp / fdl_engine.hip (tiled_mac): the workgroups of a two-region plan in one grid -- every (region, stream, tile, split,
    bin group) is decoded by exactly one workgroup, and the second region's workgroups are the last ones dispatched."""
    from kernel_model import mfma_one_grid_decode, mfma_plan
    plan = mfma_plan(T, S * groups, P, wgs)
    sizes = [S * l[1] * l[2] * groups for l in plan]
    grid = sum(sizes)
    seen = {}
    for b in range(grid):
        key = mfma_one_grid_decode(b, grid, plan, S, groups)
        assert key not in seen
        seen[key] = b
        assert key[0] == (1 if len(plan) == 2 and b >= sizes[0] else 0)
    want = {(r, s, t, q, g) for r, l in enumerate(plan) for s in range(S) for t in range(l[0], l[0] + l[1]) for q in range(l[2]) for g in range(groups)}
    assert set(seen) == want

