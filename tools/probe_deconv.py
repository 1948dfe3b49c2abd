import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, irbaboon_amd as ib
from irbaboon_amd.synth import exp_sine_sweep
N = 1 << 20
sweep = exp_sine_sweep(3 * N // 4, 48000, 20.0, 24000, fade_from=22000, n_total=N)
cap = np.roll(sweep, 100) * 0.5
for i in range(4):
    t0 = time.perf_counter(); ib.deconvolve(cap, sweep, 48000, False); print("no smoothing call", i, (time.perf_counter() - t0) * 1e3, "ms")
t0 = time.perf_counter(); ib.deconvolve(cap, sweep, 48000, True); print("smoothing", (time.perf_counter() - t0) * 1e3, "ms")
F = ib.fftTransform(cap)
t0 = time.perf_counter(); ib.averagingFilter(F, 1 / 13, 48000.0); print("one averagingFilter pass", (time.perf_counter() - t0) * 1e3, "ms")
