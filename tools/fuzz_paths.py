"""Randomised campaign over the engine's step / call paths and the one-shot function (not part of the test suite: minutes of GPU time).
Random engine shapes, IR lengths, call sizes; every output against float64 FFT convolution.
HALF of the draws run with no override at all (what a user gets); the other half pick the library's internal knobs at random
(IRB_DEBUG, irb_env.hpp).  One-shot draws: the default staged form, irb_host_alloc buffers (direct form) or the opt-in registered form.
A one-shot failure is KEPT: inputs, result, float64 reference, buffer addresses and -- if the call ran with input_checksums=1 -- the
per-block checksums of the samples the device read next to the host's, in gpurun_out/fuzz_failure_<n>.npz, so that one event is
enough to tell a stale page (device checksums differ from the host's for whole pages of input) from a kernel-side error (they agree).
usage: python tools/fuzz_paths.py [seconds] [seed]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.signal
from irbaboon_amd import FdlEngine, _lib
from irbaboon_amd._lib import planar_ptrs
from irbaboon_amd.engine import pinned_empty
from irbaboon_amd.synth import decaying_ir, white_noise

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, n, worst, t_said = time.time(), 0, 0.0, time.time()
counts = {}
lib = _lib.lib()
lib.irb_debug_input_checksums_.restype = C.c_int
lib.irb_debug_input_checksums_.argtypes = [C.POINTER(C.c_uint), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(OUT, exist_ok=True)


def set_debug(kv):
    if kv:
        os.environ["IRB_DEBUG"] = ",".join(f"{k}={v}" for k, v in kv.items())
    else:
        os.environ.pop("IRB_DEBUG", None)


def host_checksums(x, B, steps):
    """what k_push_fft_batch(_t) adds up per block: the bit patterns of its samples (zero padding adds nothing), mod 2^32"""
    out = np.zeros((x.shape[0], steps), np.uint32)
    for c in range(x.shape[0]):
        u = x[c].view(np.uint32).astype(np.uint64)
        full = x.shape[1] // B
        if full:
            out[c, :full] = (u[:full * B].reshape(full, B).sum(axis=1) & 0xFFFFFFFF).astype(np.uint32)
        if x.shape[1] % B and full < steps:
            out[c, full] = np.uint32(int(u[full * B:].sum()) & 0xFFFFFFFF)
    return out


def one_shot_draw(n):
    Bo = int(rng.choice([8, 16, 32, 100, 128, 256, 512]))
    Ca, Ci = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    La, Lho = int(rng.integers(1, 300 * Bo)), int(rng.integers(1, 40 * Bo))
    form = str(rng.choice(["staged", "staged", "pinned", "registered"]))
    dbg = {"input_checksums": 1}
    if rng.integers(0, 2):                                   # half: internal knobs at random
        dbg["mfma_min_t"] = int(rng.choice([0, 2, 256]))
        if rng.integers(0, 3) == 0:
            dbg["offline_max_blocks"] = int(rng.choice([5, 40, 700]))
        if rng.integers(0, 3) == 0:
            dbg["offline_chunk"] = int(rng.choice([16, 64, 1024]))
        dbg["offline_helpers_from"] = int(rng.choice([0, 1 << 20]))
        dbg["offline_in_one_launch"] = int(rng.integers(0, 2))
        os.environ["IRB_OFFLINE_THREADS"] = str(rng.choice([1, 2, 3, 5]))
    else:
        os.environ.pop("IRB_OFFLINE_THREADS", None)
    set_debug(dbg)
    if form == "registered":
        os.environ["IRB_OFFLINE_DIRECT"] = "1"
    else:
        os.environ.pop("IRB_OFFLINE_DIRECT", None)
    keep = bool(rng.integers(0, 2))
    xa = np.stack([white_noise(int(rng.integers(1 << 30)), La) for _ in range(Ca)])
    ha = np.stack([decaying_ir(int(rng.integers(1 << 30)), Lho) for _ in range(Ci)])
    yo = np.empty((Ca, La + Lho - 1), np.float32)
    if form == "pinned":
        px, ph, yo = pinned_empty(xa.shape), pinned_empty(ha.shape), pinned_empty(yo.shape)
        px[:], ph[:] = xa, ha
        bufs = (px, ph, yo)
    else:
        bufs = (xa, ha, yo)
    rc = lib.irb_convolve_periodic(0, planar_ptrs(bufs[0]), Ca, La, planar_ptrs(bufs[1]), Ci, Lho, Bo, 1 if keep else 0, planar_ptrs(bufs[2]))
    assert rc == 0, (lib.irb_last_error(), form, Bo, Ca, Ci, La, Lho, keep, dbg, os.environ.get("IRB_OFFLINE_THREADS"))
    prof = _lib.ConvolveProfile()
    lib.irb_convolve_last_profile(C.byref(prof))
    assert prof.direct == (0 if form == "staged" else 1), (form, prof.direct)
    yo = np.asarray(yo)
    nw = min(La + Lho - 1, (La // Bo + -(-Lho // Bo)) * Bo) if keep else La + Lho - 1
    global worst
    for c in range(Ca):
        hh = ((ha[0].astype(np.float32) + ha[1]) / np.float32(2)).astype(np.float64) if (Ci == 2 and Ca == 1) else ha[c if Ci == Ca else 0].astype(np.float64)
        ref = scipy.signal.fftconvolve(xa[c].astype(np.float64), hh)
        den = np.sqrt(np.mean(ref[:nw] ** 2))
        err = np.sqrt(np.mean((yo[c, :nw] - ref[:nw]) ** 2)) / den if den > 0 else np.abs(yo[c, :nw]).max()
        worst = max(worst, err)
        if not (err < 1e-5 and not yo[c, nw:].any()):
            # ---- keep everything that can say WHAT went wrong
            d = np.abs(yo[c, :nw].astype(np.float64) - ref[:nw])
            blocks = [(j, float(d[j * Bo:(j + 1) * Bo].max())) for j in range((nw + Bo - 1) // Bo)]
            ns, nb = C.c_int(0), C.c_int(0)
            steps = int(prof.steps)
            dev = np.zeros((Ca, steps), np.uint32)
            have_sums = lib.irb_debug_input_checksums_(dev.ctypes.data_as(C.POINTER(C.c_uint)), dev.size, C.byref(ns), C.byref(nb)) == 0
            host = host_checksums(np.asarray(bufs[0]), Bo, steps)
            bad_in = [(cc, j) for cc in range(Ca) for j in range(steps) if have_sums and dev[cc, j] != host[cc, j]]
            path = os.path.join(OUT, f"fuzz_failure_{n}.npz")
            np.savez_compressed(path, audio=xa, ir=ha, result=yo, reference_f64_channel=ref, channel=c, block=Bo, keep_reference_tail=keep,
                                form=form, debug=str(dbg), profile=str(prof.as_dict()), device_input_checksums=dev, host_input_checksums=host,
                                audio_address=np.asarray(bufs[0]).ctypes.data, ir_address=np.asarray(bufs[1]).ctypes.data, result_address=yo.ctypes.data)
            raise AssertionError(("one-shot", n, form, Bo, Ca, Ci, La, Lho, keep, dbg, "threads", os.environ.get("IRB_OFFLINE_THREADS"),
                                  float(err), "channel", c, "kept in", path, "audio base", hex(np.asarray(bufs[0]).ctypes.data),
                                  "input blocks whose device checksum differs from the host's" if have_sums else "no checksums", bad_in[:40],
                                  "output blocks off by > 1e-3 of the peak", [(j, round(v, 5)) for j, v in blocks if v > 1e-3 * np.abs(ref).max()][:40]))
    lib.irb_release_cached_plans()
    counts["one-shot " + form] = counts.get("one-shot " + form, 0) + 1


while time.time() - t0 < budget:
    for k in ("IRB_DEBUG", "IRB_OFFLINE_DIRECT", "IRB_OFFLINE_THREADS"):
        os.environ.pop(k, None)
    if n % 4 == 3:
        one_shot_draw(n)
        for k in ("IRB_DEBUG", "IRB_OFFLINE_DIRECT", "IRB_OFFLINE_THREADS"):
            os.environ.pop(k, None)
    I = int(rng.choice([1, 1, 2, 3, 5, 9, 17, 40]))
    Cc = int(rng.integers(1, 3))
    B = int(rng.choice([8, 16, 32, 64, 100, 128, 256, 512, 1024, 2048]))
    cap = int(rng.integers(1, 40)) * B + int(rng.integers(0, B))
    Lh = int(rng.integers(1, cap + 1))
    tmax = int(rng.choice([1, 1, 1, 3, 8, 20, 40, 300]))
    if tmax == 300:                                                # calls that span two tiles of the matrix-core MAC: keep the case small
        I, B = min(I, 3), min(B, 128)
        cap = int(rng.integers(1, 40)) * B + int(rng.integers(0, B))
        Lh = int(rng.integers(1, cap + 1))
    T = int(rng.integers(1, 14)) if tmax <= 8 else int(rng.integers(tmax // 2, 2 * tmax + 3))
    defaults = bool(rng.integers(0, 2))                            # half of the engines: flags 0, nothing overridden
    flags, dbg = 0, {}
    if not defaults:
        flags = int(rng.choice([0, _lib.FORCE_COLS, _lib.FORCE_COLS, _lib.FORCE_FUSED]))
        if flags == _lib.FORCE_COLS:
            dbg.update(colw=int(rng.choice([4, 16, 64])), cols_handoff=int(rng.choice([0, 2])), psplits=int(rng.integers(1, 17)))
        dbg.update(fused_u=int(rng.choice([2, 4])), mfma_min_t=int(rng.choice([0, 2, 2, 256])), mfma_wgs=int(rng.choice([16, 256, 2048])),
                   mfma_min_chunks=int(rng.choice([1, 6])), tiled_fused8=int(rng.choice([0, 1, 1])))
    set_debug(dbg)
    ir_ch = int(rng.integers(1, 3))
    x = np.stack([white_noise(int(rng.integers(1 << 30)), T * B) for _ in range(I * Cc)])
    irs = [np.stack([decaying_ir(int(rng.integers(1 << 30)), Lh) for _ in range(ir_ch)]) for _ in range(I)]
    with FdlEngine(I, Cc, B, cap, max_blocks_per_call=tmax, flags=flags) as eng:
        for i in range(I):
            eng.set_ir(i, irs[i])
        # split the T blocks into random calls
        y, j = [], 0
        while j < T:
            nb = int(rng.integers(1, T - j + 1))
            y.append(eng.process(x[:, j * B:(j + nb) * B]))
            j += nb
        y = np.concatenate(y, axis=1)
        key = ("defaults" if defaults else f"flags {flags}", eng.mac_kernel_name(min(T, tmax)))
    counts[key] = counts.get(key, 0) + 1
    for i in range(I):
        for c in range(Cc):
            if ir_ch == 2 and Cc == 1:
                h = ((irs[i][0].astype(np.float32) + irs[i][1]) / np.float32(2)).astype(np.float64)
            else:
                h = irs[i][c if (ir_ch == Cc) else 0].astype(np.float64)
            ref = scipy.signal.fftconvolve(x[i * Cc + c].astype(np.float64), h)[:T * B]
            den = np.sqrt(np.mean(ref ** 2))
            err = np.sqrt(np.mean((y[i * Cc + c] - ref) ** 2)) / den if den > 0 else np.abs(y[i * Cc + c]).max()
            worst = max(worst, err)
            if not err < 1e-5:
                path = os.path.join(OUT, f"fuzz_failure_{n}.npz")
                np.savez_compressed(path, audio=x, irs=np.stack(irs), result=y, shape=np.array([I, Cc, B, cap, Lh, T, tmax, flags]), debug=str(dbg))
                raise AssertionError(("engine", I, Cc, B, cap, Lh, T, tmax, flags, dbg, i, c, float(err), "kept in", path))
    n += 1
    if time.time() - t_said > 60:                  # (a run that stays silent for minutes is taken to be hung)
        t_said = time.time()
        print(f"... {n} engines, worst so far {worst:.2e}", flush=True)
print(f"{n} random engines in {time.time()-t0:.0f} s, worst rel RMS {worst:.2e}; by (selection, kernel): {counts}")
