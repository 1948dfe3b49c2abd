"""Measured relative RMS error of the GPU path per BASELINE configuration, against float64 FFT convolution
(scipy) and, where it finishes in seconds, the CPU oracle.  Writes gpurun_out/accuracy.txt when run on the box (committed as profiles/rNN_accuracy.txt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.signal
import irbaboon_amd as ib, oracle
from irbaboon_amd.synth import decaying_ir, white_noise
from util import rel_rms

rows = []
def case(name, C, La, Lh, B, with_oracle):
    x = np.stack([white_noise(1000 + c, La) for c in range(C)]); h = np.stack([decaying_ir(9000 + c, Lh) for c in range(C)])
    y = ib.convolvePeriodic(x, h, B, keep_reference_tail=False)
    e64 = max(rel_rms(y[c], scipy.signal.fftconvolve(x[c].astype(np.float64), h[c].astype(np.float64))) for c in range(C))
    eo = None
    if with_oracle:
        r, _ = oracle.convolve_periodic(x, h, B)
        n = (La // B + int(np.ceil(Lh / B))) * B
        eo = max(rel_rms(y[c, :n], r[c, :n]) for c in range(C))
        eoo = max(rel_rms(r[c, :n], scipy.signal.fftconvolve(x[c].astype(np.float64), h[c].astype(np.float64))[:n]) for c in range(C))
    rows.append(f"{name:<58} GPU vs float64: {e64:.2e}" + (f"   GPU vs oracle: {eo:.2e}   oracle vs float64: {eoo:.2e}" if eo is not None else ""))

case("configs[0] mono B=256 Lh=4096 La=480000 (offline call)", 1, 480000, 4096, 256, True)
case("configs[1] mono B=256 Lh=96000 La=960000 (offline call)", 1, 960000, 96000, 256, True)
case("configs[2] stereo B=128 Lh=480000 La=512000 (offline call)", 2, 512000, 480000, 128, False)
# streaming engine (one block per call): configs[1] and configs[3] geometry, 40 blocks
for name, B, Lh, flags in (("configs[1] streaming, fused kernel, 40 blocks", 256, 96000, 1), ("configs[1] streaming, split path, 40 blocks", 256, 96000, 2),
                           ("configs[3] geometry B=512 streaming, fused, 40 blocks", 512, 96000, 1)):
    x = white_noise(5, 40 * B)[None]; h = decaying_ir(6, Lh)
    with ib.FdlEngine(1, 1, B, Lh, flags=flags) as e:
        e.set_ir(0, h); y = e.process(x)
    rows.append(f"{name:<58} GPU vs float64: {rel_rms(y[0], scipy.signal.fftconvolve(x[0].astype(np.float64), h.astype(np.float64))[:40 * B]):.2e}")
out = "\n".join(["# relative RMS error, tolerance 1e-5 (north star); seeded white noise, decaying-noise IRs of unit norm"] + rows)
print(out)
open(os.path.join(ROOT, "gpurun_out", "accuracy.txt"), "w").write(out + "\n")
