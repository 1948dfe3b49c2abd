"""Ad-hoc soak: tens of thousands of streaming steps (ring wraps thousands of times) on every step path;
the last blocks are checked against float64 FFT convolution."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.signal
from irbaboon_amd import FdlEngine, _lib
from irbaboon_amd.synth import decaying_ir, white_noise

B, Lh, I, C, steps, chunk = 64, 1000, 6, 2, 20000, 500
x = np.stack([white_noise(1 + s, steps * B) for s in range(I * C)])
irs = [np.stack([decaying_ir(50 + 2 * i + c, Lh) for c in range(C if i % 2 else 1)]) for i in range(I)]
paths = [("fused(+pair)", _lib.FORCE_FUSED, 1, None), ("tiled", 0, 7, None), ("tiled mfma", 0, 500, None),
         ("cols ticket", _lib.FORCE_COLS, 1, "0"), ("cols granules", _lib.FORCE_COLS, 1, "2")]
for name, flags, tmax, handoff in paths:
    dbg = ["mfma_min_t=" + ("256" if name == "tiled mfma" else "0")]       # calls of 500 blocks on the matrix cores / never
    if handoff is not None:
        dbg.append("cols_handoff=" + handoff)                               # read when the engine is created
    os.environ["IRB_DEBUG"] = ",".join(dbg)
    t0 = time.time()
    with FdlEngine(I, C, B, Lh, max_blocks_per_call=tmax, flags=flags) as eng:
        for i in range(I):
            eng.set_ir(i, irs[i])
        y = np.concatenate([eng.process(x[:, k:k + chunk * B]) for k in range(0, steps * B, chunk * B)], axis=1)
    worst = 0.0
    tail = 200 * B
    for i in range(I):
        for c in range(C):
            h = irs[i][c if irs[i].shape[0] == C else 0].astype(np.float64)
            ref = scipy.signal.fftconvolve(x[i * C + c].astype(np.float64), h)[steps * B - tail:steps * B]
            got = y[i * C + c, -tail:]
            worst = max(worst, float(np.sqrt(np.mean((got - ref) ** 2) / np.mean(ref ** 2))))
    print(f"{name:15s} {steps} steps, ring {eng.info.ring_slots} slots: worst rel RMS over the last 200 blocks {worst:.2e}  ({time.time()-t0:.1f} s)")
    assert worst < 1e-5
