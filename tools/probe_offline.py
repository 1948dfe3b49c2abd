import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, irbaboon_amd as ib
from irbaboon_amd.synth import decaying_ir, white_noise
import oracle
for (La, Lh, B, C) in [(960000, 96000, 256, 1), (480000, 4096, 256, 1), (128 * 4000, 480000, 128, 2)]:
    x = np.stack([white_noise(1 + c, La) for c in range(C)]); h = np.stack([decaying_ir(9 + c, Lh) for c in range(C)])
    for i in range(3):
        t0 = time.perf_counter(); y = ib.convolvePeriodic(x, h, B); dt = time.perf_counter() - t0
        print(f"convolvePeriodic La={La} Lh={Lh} B={B} C={C} call {i}: {dt*1e3:.2f} ms  ({C*La/dt/1e6:.1f} Msamples/s end to end)")
    if Lh <= 96000:
        t0 = time.perf_counter(); oracle.convolve_periodic(x, h, B); dt = time.perf_counter() - t0
        print(f"   oracle: {dt*1e3:.1f} ms")
