"""The stale-page probe (round 4's first form -- identical content every repeat -- could not see a stale page that still held the previous
repeat's bytes and was removed): every call has its own random lengths (stereo, B = 256), large float64 temporaries are
allocated and freed between calls (what scipy's reference convolution does in tools/fuzz_paths.py: mmap / munmap churn), and the
opt-in REGISTERED form's result (IRB_OFFLINE_DIRECT=1) is compared bit for bit with the default STAGED form's (read per call; same
kernels, the samples travel through the library's pinned memory instead of the registered caller memory).  Not run in round 5
(no more GPU minutes on reproducing the event; the registered form is no longer a default).
usage: python tools/stale_page_probe2.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from irbaboon_amd import _lib
from irbaboon_amd._lib import planar_ptrs
from irbaboon_amd.synth import decaying_ir, white_noise

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
os.environ["IRB_DEBUG"] = "mfma_min_t=2,mfma_wgs=2048,mfma_min_chunks=1"
lib = _lib.lib()
B = 256
noise = [white_noise(100 + k, 80000) for k in range(8)]
irs = [decaying_ir(200 + k, 10240) for k in range(8)]
t_end = time.time() + seconds
calls = events = 0
while time.time() < t_end:
    La, Lh = int(rng.integers(2000, 76800)), int(rng.integers(300, 10240))
    x = np.stack([noise[int(rng.integers(8))][:La], noise[int(rng.integers(8))][:La]])
    h = np.stack([irs[int(rng.integers(8))][:Lh], irs[int(rng.integers(8))][:Lh]])
    res = []
    for direct in ("1", None):
        if direct:
            os.environ["IRB_OFFLINE_DIRECT"] = direct
        else:
            os.environ.pop("IRB_OFFLINE_DIRECT", None)
        out = np.full((2, La + Lh - 1), 1e30, np.float32)
        assert lib.irb_convolve_periodic(0, planar_ptrs(x), 2, La, planar_ptrs(h), 2, Lh, B, 1, planar_ptrs(out)) == 0
        res.append(out)
        tmp = [np.zeros(int(rng.integers(100000, 2000000)), np.float64) for _ in range(3)]     # churn: mmap'd, touched, freed
        del tmp
    calls += 1
    if not np.array_equal(res[0], res[1]):
        events += 1
        for c in range(2):
            d = np.flatnonzero(res[0][c] != res[1][c])
            if d.size:
                s0 = int(d[0])
                addr = x[c].ctypes.data + 4 * s0
                print(f"EVENT after {calls} calls: La={La} Lh={Lh} channel {c}: {d.size} samples differ, first {s0} (block {s0 // B}), last {int(d[-1])}; input address of the "
                      f"first differing sample {addr:#x} = page offset {addr % 4096}; x base {x.ctypes.data:#x}", flush=True)
    if calls % 5 == 0:
        lib.irb_release_cached_plans()
    if calls % 5000 == 0:
        print(f"... {calls} calls, {events} events", flush=True)
print(f"{calls} pairs of one-shot calls (direct against staged, stereo, B = 256, random lengths, allocation churn between them), {events} with differing bits")
