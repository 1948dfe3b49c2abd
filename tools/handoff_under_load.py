"""Granule hand-off under the heaviest co-tenant this repo has: the 2048-instance fused engine (every CU full, HBM
saturated) streams on its own HIP stream while a one-stream and a stereo long-IR engine take thousands of one-block
steps (k_step_cols, tagged granules).  Checks: no hand-off timeout (IRB_ERR_STATE), output equal to the fused kernel's."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from irbaboon_amd import FdlEngine, _lib
from irbaboon_amd.synth import decaying_ir, white_noise

stop = threading.Event()
def background():
    with FdlEngine(2048, 1, 256, 96000, flags=_lib.FORCE_FUSED) as bg:
        h = decaying_ir(1, 96000)
        for i in range(0, 2048, 8):
            bg.set_ir(i, h)
        d_in, d_out, stride = bg.device_io()
        while not stop.is_set():
            for _ in range(10):
                bg.process_device(d_in, d_out, stride, 1)
            bg.synchronize()

th = threading.Thread(target=background); th.start()
time.sleep(3.0)
try:
    for I, C, B, Lh, T in ((1, 1, 256, 96000, 1500), (1, 2, 128, 480000, 1500), (8, 1, 256, 24000, 800)):
        x = np.stack([white_noise(10 + s, T * B) for s in range(I * C)])
        irs = [np.stack([decaying_ir(100 + 2 * i + c, Lh) for c in range(C)]) for i in range(I)]
        outs, times = [], []
        for flags in (_lib.FORCE_COLS, _lib.FORCE_FUSED):
            with FdlEngine(I, C, B, Lh, flags=flags) as eng:
                for i in range(I):
                    eng.set_ir(i, irs[i])
                t0 = time.time()
                y = np.concatenate([eng.process(x[:, j * B:(j + 1) * B]) for j in range(T)], axis=1)
                times.append((time.time() - t0) / T * 1e6)
            outs.append(y)
        err = max(float(np.sqrt(np.mean((outs[0][s] - outs[1][s]) ** 2) / np.mean(outs[1][s] ** 2))) for s in range(I * C))
        print(f"I={I} C={C} B={B} Lh={Lh}: {T} steps under load, cols (granules) vs fused rel RMS {err:.2e}; "
              f"host-API step {times[0]:.0f} us (cols) / {times[1]:.0f} us (fused)")
        assert err < 2e-6
finally:
    stop.set(); th.join()
print("ok: no hand-off timeout")
