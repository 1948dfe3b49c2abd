"""Looks for the rare wrong one-shot result (profiles/NOTES.md: stale input pages behind the registered mapping) without a reference
computation: the SAME stereo call again and again from freshly allocated arrays of identical content, results compared bit for bit
with the first.  On a mismatch: which samples differ, and where the first differing sample's INPUT sample lies in its page (the first
wrong output sample of a causal convolution whose first tap is not zero is the first wrong input sample).
usage: python tools/stale_page_probe.py [seconds] [seed] [rebuild] [mfma] [engines]     rebuild = 1: the thread's cached engine is freed after every call (pinned and
device memory churn between the registrations, as in tools/fuzz_paths.py)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from irbaboon_amd import FdlEngine, _lib
from irbaboon_amd._lib import planar_ptrs
from irbaboon_amd.synth import decaying_ir, white_noise

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = _lib.lib()
rebuild = len(sys.argv) > 3 and sys.argv[3] == "1"
if len(sys.argv) > 4 and sys.argv[4] == "mfma":           # the matrix-core path forced onto these short calls, as in both failing campaign calls
    os.environ.update({"IRB_TILED_MFMA_MIN_T": "2", "IRB_TILED_MFMA_WGS": "2048", "IRB_TILED_MFMA_MIN_CHUNKS": "1"})
engines = len(sys.argv) > 5 and sys.argv[5] == "engines"   # an engine of a random shape created, used and destroyed between two calls (the campaign does three)
B = 256
t_end = time.time() + seconds
calls = events = 0
junk = []
while time.time() < t_end:
    La, Lh = int(rng.integers(20000, 60000)), int(rng.integers(3000, 9000))
    x0 = np.stack([white_noise(int(rng.integers(1 << 30)), La) for _ in range(2)])
    h0 = np.stack([decaying_ir(int(rng.integers(1 << 30)), Lh) for _ in range(2)])
    h0[:, 0] = 1.0                                        # a first tap that is certainly not zero
    first = None
    for rep in range(200):
        junk.append(np.empty(int(rng.integers(1, 200000)), np.uint8))      # heap churn: the arrays below land at varying addresses
        if len(junk) > 40:
            del junk[:int(rng.integers(1, 30))]
        x, h = x0.copy(), h0.copy()
        out = np.full((2, La + Lh - 1), 1e30, np.float32)
        assert lib.irb_convolve_periodic(0, planar_ptrs(x), 2, La, planar_ptrs(h), 2, Lh, B, 1, planar_ptrs(out)) == 0
        calls += 1
        if engines:
            eb, ei, ec = int(rng.choice([8, 32, 128, 256])), int(rng.integers(1, 6)), int(rng.integers(1, 3))
            with FdlEngine(ei, ec, eb, int(rng.integers(3, 30)) * eb, max_blocks_per_call=int(rng.choice([1, 3, 8, 40, 300])),
                           flags=int(rng.choice([0, _lib.FORCE_COLS, _lib.FORCE_FUSED, _lib.FORCE_SPLIT]))) as eng:
                for i in range(ei):
                    eng.set_ir(i, decaying_ir(7 + i, eb * 3)[None])
                eng.process(np.stack([white_noise(3 + k, eb) for k in range(ei * ec)]))
        if rebuild:
            lib.irb_release_cached_plans()
        if first is None:
            first = out
            continue
        if not np.array_equal(first, out):
            events += 1
            for c in range(2):
                d = np.flatnonzero(first[c] != out[c])
                if d.size == 0:
                    continue
                s0, s1 = int(d[0]), int(d[-1])
                addr = x[c].ctypes.data + 4 * s0
                big = np.flatnonzero(np.abs(first[c] - out[c]) > 1e-3 * np.abs(first[c]).max())
                print(f"EVENT after {calls} calls: La={La} Lh={Lh} channel {c}: {d.size} samples differ, first {s0} (block {s0 // B}), last {s1}; "
                      f"samples off by > 1e-3 of the peak: {big.size}, from {int(big[0]) if big.size else -1} to {int(big[-1]) if big.size else -1}; "
                      f"input address of the first differing sample {addr:#x} = page offset {addr % 4096} bytes; input array base {x.ctypes.data:#x}, "
                      f"IR base {h.ctypes.data:#x}, result base {out.ctypes.data:#x}; input unchanged on the CPU side: {bool(np.array_equal(x, x0))}", flush=True)
    if calls % 10000 < 200:
        print(f"... {calls} calls, {events} events", flush=True)
print(f"{calls} one-shot calls (stereo, B = 256, fresh arrays every call{', engine rebuilt every call' if rebuild else ''}), {events} with bits differing from the first call of their shape")
