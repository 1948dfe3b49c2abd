"""Run-to-run determinism of the one-shot function and of engine calls: the same call repeated must give the same BITS (no
atomics or arrival-order sums anywhere on these paths; a race would show here long before it shows against a tolerance).
usage: python tools/determinism_probe.py [seed] [shapes] [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from irbaboon_amd import FdlEngine, _lib
from irbaboon_amd._lib import planar_ptrs
from irbaboon_amd.synth import decaying_ir, white_noise

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shapes = int(sys.argv[2]) if len(sys.argv) > 2 else 30
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rng = np.random.default_rng(seed)
lib = _lib.lib()
bad = 0
for k in range(shapes):
    B = int(rng.choice([32, 128, 256, 256, 512]))
    Ca, Ci = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    La, Lh = int(rng.integers(B, 400 * B)), int(rng.integers(1, 60 * B))
    for key in list(os.environ):
        if key.startswith("IRB_"):
            os.environ.pop(key)
    env = {"IRB_DEBUG": "mfma_min_t=" + str(rng.choice([2, 256]))}
    if rng.integers(0, 3) == 0:
        env["IRB_OFFLINE_DIRECT"] = "1"                  # the opt-in registered form; otherwise the default (staged) form
    os.environ.update(env)
    x = np.stack([white_noise(int(rng.integers(1 << 30)), La) for _ in range(Ca)])
    h = np.stack([decaying_ir(int(rng.integers(1 << 30)), Lh) for _ in range(Ci)])
    first = None
    for r in range(reps):
        out = np.full((Ca, La + Lh - 1), 1e30, np.float32)
        assert lib.irb_convolve_periodic(0, planar_ptrs(x), Ca, La, planar_ptrs(h), Ci, Lh, B, int(r % 2 == 0 or True), planar_ptrs(out)) == 0
        if first is None:
            first = out
        elif not np.array_equal(first, out):
            bad += 1
            d = np.argwhere(first != out)
            print("DIFFERENT BITS: one-shot", B, Ca, Ci, La, Lh, env, "repeat", r, len(d), "samples, first at", d[0].tolist(), "blocks", sorted({int(i[1]) // B for i in d})[:20], flush=True)
            break
        if r % 7 == 3:
            lib.irb_release_cached_plans()            # (a rebuilt engine: other device addresses)
    lib.irb_release_cached_plans()
    # the same through an engine, device-resident, T blocks per call
    T = int(rng.choice([8, 40, 300, 700]))
    P = int(rng.integers(1, 40))
    xs = np.stack([white_noise(int(rng.integers(1 << 30)), 3 * T * B) for _ in range(Ca)])
    with FdlEngine(1, Ca, B, P * B, max_blocks_per_call=T) as eng:
        eng.set_ir(0, np.stack([decaying_ir(int(rng.integers(1 << 30)), P * B) for _ in range(Ci)]))
        first = None
        for r in range(reps // 4):
            eng.reset()
            y = eng.process(xs)
            if first is None:
                first = y
            elif not np.array_equal(first, y):
                bad += 1
                print("DIFFERENT BITS: engine", B, Ca, Ci, T, P, env, eng.mac_kernel_name(T), "repeat", r, flush=True)
                break
    print(f"... shape {k}: B={B} {Ca}x{Ci} La={La} Lh={Lh} {env} ok", flush=True)
print(f"{shapes} shapes x {reps} one-shot repeats (+ {reps // 4} engine repeats each): {bad} with differing bits")
