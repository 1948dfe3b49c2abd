"""Randomised campaign over every one-block step path and its knobs (not part of the test suite: minutes of GPU time).
Random engine shapes, IR lengths, call sizes, hand-off forms and slice widths; every output against float64 FFT convolution.
usage: python tools/fuzz_paths.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.signal
from irbaboon_amd import FdlEngine, _lib
from irbaboon_amd.synth import decaying_ir, white_noise

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, n, worst, t_said = time.time(), 0, 0.0, time.time()
counts = {}
while time.time() - t0 < budget:
    I = int(rng.choice([1, 1, 2, 3, 5, 9, 17, 40]))
    C = int(rng.integers(1, 3))
    B = int(rng.choice([8, 16, 32, 64, 100, 128, 256, 512, 1024, 2048]))
    cap = int(rng.integers(1, 40)) * B + int(rng.integers(0, B))
    Lh = int(rng.integers(1, cap + 1))
    T = int(rng.integers(1, 14))
    tmax = int(rng.choice([1, 1, 1, 3, 8, 20, 40, 300]))
    if tmax == 300:                                                # calls that span two tiles of the matrix-core MAC: keep the case small
        I, B = min(I, 3), min(B, 128)
        cap = int(rng.integers(1, 40)) * B + int(rng.integers(0, B))
        Lh = int(rng.integers(1, cap + 1))
    T = int(rng.integers(1, 14)) if tmax <= 8 else int(rng.integers(tmax // 2, 2 * tmax + 3))
    flags = int(rng.choice([0, _lib.FORCE_COLS, _lib.FORCE_COLS, _lib.FORCE_FUSED, _lib.FORCE_SPLIT, _lib.FORCE_SPLIT | _lib.SPLIT_3LAUNCH]))
    env = {}
    if flags == _lib.FORCE_COLS:
        env = {"IRB_COLW": str(rng.choice([4, 16, 64])), "IRB_COLS_HANDOFF": str(rng.choice([0, 1, 2])), "IRB_PSPLITS": str(rng.integers(1, 17))}
    env["IRB_TILED_KERNEL"] = str(rng.choice([0, 1, 1]))        # round 2's tiled MAC / the packed-FMA one
    env["IRB_FUSED_U"] = str(rng.choice([2, 4]))
    env["IRB_TILED_MFMA_MIN_T"] = str(rng.choice([0, 2, 2, 256]))    # off / the matrix-core tiled MAC for every call of >= 2 blocks / the default rule
    env["IRB_TILED_MFMA_WGS"] = str(rng.choice([16, 256, 2048]))
    env["IRB_TILED_MFMA_MIN_CHUNKS"] = str(rng.choice([1, 6]))
    env["IRB_TILED_MFMA_TWO_LAUNCH"] = str(rng.choice([0, 1, 1]))   # whole rounds + remainder launch of the matrix-core MAC
    env["IRB_TILED_MFMA_RES"] = str(rng.choice([1, 1, 2]))
    env["IRB_TILED_MFMA_ONE_GRID"] = str(rng.choice([0, 1, 1]))    # both regions of a plan in one grid / two launches
    env["IRB_TILED_FUSED8"] = str(rng.choice([0, 1, 1]))            # calls of <= 8 blocks as one launch / three
    for k in ("IRB_COLW", "IRB_COLS_HANDOFF", "IRB_PSPLITS", "IRB_TILED_KERNEL", "IRB_FUSED_U", "IRB_TILED_MFMA_MIN_T", "IRB_TILED_MFMA_WGS", "IRB_TILED_MFMA_MIN_CHUNKS",
              "IRB_TILED_MFMA_TWO_LAUNCH", "IRB_TILED_MFMA_RES", "IRB_TILED_MFMA_ONE_GRID", "IRB_TILED_FUSED8", "IRB_OFFLINE_DIRECT", "IRB_OFFLINE_MAX_BLOCKS", "IRB_OFFLINE_CHUNK", "IRB_OFFLINE_HELPER"):
        os.environ.pop(k, None)
    os.environ.update(env)
    if n % 4 == 3:
        # the one-shot function (irb_convolve_periodic) in both forms, in one or several pieces, with and without the reference's dropped tail
        from irbaboon_amd import convolvePeriodic
        Bo = int(rng.choice([8, 16, 32, 100, 128, 256, 512]))
        Ca, Ci = int(rng.integers(1, 3)), int(rng.integers(1, 3))
        La, Lho = int(rng.integers(1, 300 * Bo)), int(rng.integers(1, 40 * Bo))
        oenv = {"IRB_OFFLINE_DIRECT": str(rng.choice([0, 1, 1])), "IRB_OFFLINE_HELPER": str(rng.choice([0, 1])), "IRB_OFFLINE_CHUNK": str(rng.choice([16, 64, 1024]))}
        if rng.integers(0, 3) == 0:
            oenv["IRB_OFFLINE_MAX_BLOCKS"] = str(rng.choice([5, 40, 700]))
        os.environ.update(oenv)
        keep = bool(rng.integers(0, 2))
        xa = np.stack([white_noise(int(rng.integers(1 << 30)), La) for _ in range(Ca)])
        ha = np.stack([decaying_ir(int(rng.integers(1 << 30)), Lho) for _ in range(Ci)])
        yo = convolvePeriodic(xa, ha, Bo, keep_reference_tail=keep)
        nw = min(La + Lho - 1, (La // Bo + -(-Lho // Bo)) * Bo) if keep else La + Lho - 1
        for c in range(Ca):
            hh = ((ha[0].astype(np.float32) + ha[1]) / np.float32(2)).astype(np.float64) if (Ci == 2 and Ca == 1) else ha[c if Ci == Ca else 0].astype(np.float64)
            ref = scipy.signal.fftconvolve(xa[c].astype(np.float64), hh)
            den = np.sqrt(np.mean(ref[:nw] ** 2))
            err = np.sqrt(np.mean((yo[c, :nw] - ref[:nw]) ** 2)) / den if den > 0 else np.abs(yo[c, :nw]).max()
            worst = max(worst, err)
            if not (err < 1e-5 and not yo[c, nw:].any()):              # which blocks are wrong (a failure here has been seen once in ~90 000 calls and not reproduced)
                d = np.abs(yo[c, :nw].astype(np.float64) - ref[:nw])
                blocks = [(j, float(d[j * Bo:(j + 1) * Bo].max())) for j in range((nw + Bo - 1) // Bo)]
                raise AssertionError(("one-shot", n, Bo, Ca, Ci, La, Lho, keep, oenv, env, err, "channel", c, "peak", float(np.abs(ref).max()),
                                      "blocks off by > 1e-3 of it", [(j, round(v, 5)) for j, v in blocks if v > 1e-3 * np.abs(ref).max()][:40]))
        _lib.lib().irb_release_cached_plans()
        counts["one-shot"] = counts.get("one-shot", 0) + 1
    ir_ch = int(rng.integers(1, 3))
    x = np.stack([white_noise(int(rng.integers(1 << 30)), T * B) for _ in range(I * C)])
    irs = [np.stack([decaying_ir(int(rng.integers(1 << 30)), Lh) for _ in range(ir_ch)]) for _ in range(I)]
    with FdlEngine(I, C, B, cap, max_blocks_per_call=tmax, flags=flags) as eng:
        for i in range(I):
            eng.set_ir(i, irs[i])
        # split the T blocks into random calls
        y, j = [], 0
        while j < T:
            nb = int(rng.integers(1, T - j + 1))
            y.append(eng.process(x[:, j * B:(j + nb) * B]))
            j += nb
        y = np.concatenate(y, axis=1)
        key = (flags, eng.info.splits > 1)
    counts[key] = counts.get(key, 0) + 1
    for i in range(I):
        for c in range(C):
            if ir_ch == 2 and C == 1:
                h = ((irs[i][0].astype(np.float32) + irs[i][1]) / np.float32(2)).astype(np.float64)
            else:
                h = irs[i][c if (ir_ch == C) else 0].astype(np.float64)
            ref = scipy.signal.fftconvolve(x[i * C + c].astype(np.float64), h)[:T * B]
            den = np.sqrt(np.mean(ref ** 2))
            err = np.sqrt(np.mean((y[i * C + c] - ref) ** 2)) / den if den > 0 else np.abs(y[i * C + c]).max()
            worst = max(worst, err)
            assert err < 1e-5, (I, C, B, cap, Lh, T, tmax, flags, env, i, c, err)
    n += 1
    if time.time() - t_said > 60:                  # (a run that stays silent for minutes is taken to be hung)
        t_said = time.time()
        print(f"... {n} engines, worst so far {worst:.2e}", flush=True)
print(f"{n} random engines in {time.time()-t0:.0f} s, worst rel RMS {worst:.2e}; by (flags, multi-workgroup): {counts}")
