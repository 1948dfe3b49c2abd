"""The drop-in function as a maintainer binds it: irb_convolve_periodic (fp::convolution::convolvePeriodic, fp/convolution.hpp:31)
from pageable host buffers, wall clock, with the library's own breakdown of the call.
usage: python tools/oneshot_probe.py [cfg0|cfg1|cfg2|all] [reps] [pinned]
Default = the staged form (pageable numpy buffers through the library's pinned staging); `pinned` puts audio, IR and result into
irb_host_alloc memory (direct form, nothing registered); IRB_OFFLINE_DIRECT=1 opts in to per-call registration of the pageable
buffers.  IRB_OFFLINE_THREADS = host threads of the staged form (default 5).  Internal knobs: IRB_DEBUG="offline_chunk=..,
offline_max_blocks=..,offline_helpers_from=..,offline_pieces=..(how many pieces a long signal is cut into at most, default 2)"."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from irbaboon_amd import _lib, convolvePeriodic
from irbaboon_amd._lib import planar_ptrs
from irbaboon_amd.synth import decaying_ir, white_noise

SHAPES = {  # name: (audio channels, IR channels, B, Lh, La)
    "cfg0": (1, 1, 256, 4096, 480000),
    "cfg1": (1, 1, 256, 96000, 960000),
    "cfg2": (2, 2, 128, 480000, 480000 * 2),
}


def one(name, reps, pinned=False):
    Ca, Ci, B, Lh, La = SHAPES[name]
    x = np.stack([white_noise(10 + c, La) for c in range(Ca)])
    h = np.stack([decaying_ir(20 + c, Lh) for c in range(Ci)])
    if pinned:
        from irbaboon_amd.engine import pinned_empty
        px, ph = pinned_empty(x.shape), pinned_empty(h.shape)
        px[:], ph[:] = x, h
        x, h = px, ph
    lib = _lib.lib()
    lib.irb_convolve_profile_events(0)
    y = convolvePeriodic(x, h, B)                      # builds the engine
    import scipy.signal
    ref = scipy.signal.fftconvolve(x[0].astype(np.float64), h[0].astype(np.float64))
    nw = (La // B + -(-Lh // B)) * B
    err = float(np.sqrt(np.mean((y[0, :nw] - ref[:nw]) ** 2) / np.mean(ref[:nw] ** 2)))
    # (a) the Python mirror: fresh output array per call, as a caller of the reference's function gets a fresh buffer
    ta = []
    for _ in range(reps):
        t0 = time.perf_counter()
        convolvePeriodic(x, h, B)
        ta.append(time.perf_counter() - t0)
    # (b) the C ABI itself with the caller's buffers reused
    if pinned:
        out = pinned_empty((Ca, La + Lh - 1))
        out[:] = 0
    else:
        out = np.zeros((Ca, La + Lh - 1), np.float32)
    xp, hp, op = planar_ptrs(x), planar_ptrs(h), planar_ptrs(out)
    tb, profs = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        rc = lib.irb_convolve_periodic(0, xp, Ca, La, hp, Ci, Lh, B, 1, op)
        tb.append(time.perf_counter() - t0)
        assert rc == 0
    # (c) the library's breakdown, device events on (those calls wait once more per piece: not the ones timed above)
    lib.irb_convolve_profile_events(1)
    for _ in range(max(5, reps // 3)):
        rc = lib.irb_convolve_periodic(0, xp, Ca, La, hp, Ci, Lh, B, 1, op)
        assert rc == 0
        p = _lib.ConvolveProfile()
        lib.irb_convolve_last_profile(C.byref(p))
        profs.append(p.as_dict())
    med = {k: float(np.median([p[k] for p in profs])) for k in profs[0]}
    env = {k: v for k, v in os.environ.items() if k.startswith("IRB_OFFLINE") or k == "IRB_DEBUG"}
    if pinned:
        env["buffers"] = "irb_host_alloc"
    print(f"{name} {env}: python mirror {np.median(ta)*1e3:.3f} ms (min {min(ta)*1e3:.3f}), C ABI {np.median(tb)*1e3:.3f} ms (min {min(tb)*1e3:.3f}); "
          f"rel RMS vs float64 {err:.2e}")
    print("    " + "  ".join(f"{k}={med[k]:.3f}" if isinstance(profs[0][k], float) else f"{k}={int(med[k])}" for k in med))
    lib.irb_release_cached_plans()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 21
    for n in (SHAPES if which == "all" else [which]):
        one(n, reps, pinned=len(sys.argv) > 3 and sys.argv[3] == "pinned")
