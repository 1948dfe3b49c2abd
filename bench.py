#!/usr/bin/env python3
"""bench.py -- partitioned-convolution throughput on MI355X.

Workload (BASELINE.json metric "Msamples/s (48 kHz, 256-blk, 2 s IR)", configs[1] geometry):
mono streams, block B=256, IR 96 000 samples (P=375 partitions, N=512), streaming model: one
step = every resident instance consumes ONE new 256-sample block (forward FFT into the FDL ring,
multiply-accumulate over all 375 partitions, inverse FFT, overlap-add).  A single such stream is
cache-resident and launch-bound (its numbers are reported under "single_stream"); the headline
replicates the configuration over `--instances` independent instances per GPU (distinct IRs) so
that the working set (1.5 MiB per instance) is far beyond the 256 MiB Infinity Cache and the
kernel is measured against HBM.  Instances are sharded over GPUs with no collective (weak scaling).

One JSON line on stdout (rank 0).  `value` counts channel-samples through the convolver with
inputs and outputs resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, LH = 256, 96000          # configs[1]: block 256, 2 s IR at 48 kHz
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming copy)


def cpu_baseline(seconds):
    """The CPU oracle (a port of the reference's convolvePeriodic loop, oracle/fp_oracle.c) timed
    on one host core on a bounded sample of the same workload: one mono stream, B=256, Lh=96000."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from irbaboon_amd.synth import decaying_ir, white_noise
    h = decaying_ir(18000, LH)[None]
    probe = white_noise(2000, LH)[None]
    oracle.convolve_periodic(probe[:, :LH // 8], h, B)            # warm up (page in the library, allocator)
    t0 = time.perf_counter()
    oracle.convolve_periodic(probe, h, B)
    per_step = (time.perf_counter() - t0) / (probe.shape[1] // B + LH // B)     # seconds per block step (audio + tail blocks)
    La = int(max(4, min(400, (seconds / per_step - LH // B) * B / LH))) * LH
    x = white_noise(2001, La)[None]
    t0 = time.perf_counter()
    oracle.convolve_periodic(x, h, B)
    dt = time.perf_counter() - t0
    return {"value": round(La / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"1 mono stream, B={B}, Lh={LH}, La={La} samples ({La // B}+375 block steps), {dt:.1f} s"}


def cpu_baseline_multicore(single, seconds):
    """One independent stream per host core (the reference is single-threaded per stream): aggregate rate.
    Threads suffice: each stream is one long C call and ctypes drops the GIL for its duration."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from irbaboon_amd.synth import decaying_ir, white_noise
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))       # a one-GPU box's CPU share is 16 cores
    La = int(max(2, min(100, seconds * single["value"] * 1e6 / LH))) * LH
    data = [(white_noise(2000 + i, La)[None], decaying_ir(18000 + i, LH)[None]) for i in range(cores)]
    oracle.lib()
    ts = [threading.Thread(target=oracle.convolve_periodic, args=(x, h, B)) for x, h in data]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    wall = time.perf_counter() - t0
    return {"value": round(cores * La / wall / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{cores} independent mono streams (one thread each), B={B}, Lh={LH}, La={La} each, {wall:.1f} s wall"}


def _timed_calls(eng, d_in, d_out, stride, T, reps):
    eng.process_device(d_in, d_out, stride, T)
    eng.synchronize()
    eng.mark(0)
    for _ in range(reps):
        eng.process_device(d_in, d_out, stride, T)
    eng.mark(1)
    return eng.elapsed_ms() / reps


def other_configs(dev, np, FdlEngine, decaying_ir, white_noise):
    """The other BASELINE.json configurations, measured the same way (device-resident audio, hipEvents);
    parity for each is covered by tests/.  Not part of `value`."""
    res = {}

    def run(name, I, C, Bk, Lh, T, reps, note, ir_ch=None):
        e = FdlEngine(I, C, Bk, Lh, max_blocks_per_call=T, device=dev)
        for i in range(I):
            e.set_ir(i, np.stack([decaying_ir(27000 + 2 * i + c, Lh) for c in range(ir_ch or C)]))
        di, do, st = e.device_io()
        e.upload(di, np.stack([white_noise(3000 + s, T * Bk) for s in range(I * C)]))
        ms = _timed_calls(e, di, do, st, T, reps)
        alg = I * C * T * e.info.algorithmic_bytes_per_block_step
        res[name] = {"Msamples_per_s": round(I * C * T * Bk / ms / 1e3, 1), "ms_per_call": round(ms, 4), "blocks_per_call": T,
                     "algorithmic_GBps_streaming_model": round(alg / ms / 1e6, 1), "note": note}
        e.close()

    run("configs[1]_offline_one_stream", 1, 1, 256, 96000, 3750, 10,
        "ONE mono stream, whole 960000-sample signal in one call (time-tiled MAC); cache-resident, compute-bound")
    run("configs[1]_x2048_8_blocks_per_call", 2048, 1, 256, 96000, 8, 10,
        "2048 instances, 8 blocks per call: H and FDL rows are read once per 8 outputs")
    run("configs[2]_stereo_10s_ir_streaming", 1, 2, 128, 480000, 1, 200,
        "stereo, B=128, P=3750, one block per call (split-P path, 14.6 MiB working set: Infinity-Cache resident)")
    run("configs[2]_stereo_10s_ir_offline", 1, 2, 128, 480000, 2048, 5, "same IRs, 2048 blocks per call (time-tiled)")
    run("configs[3]_1024_instances_b512_streaming", 1024, 1, 512, 96000, 1, 100,
        "1024 mono instances, B=512, P=188, one block per call: HBM-bound (1.5 GiB working set)")
    run("rt_layout_1024_stereo_instances_mono_ir", 1024, 2, 256, 96000, 1, 100,
        "stereo audio, IR channel 0 for both channels (the plug-in's real-time layout, Source/PluginProcessor.cpp:485): "
        "paired kernel loads each IR row once for both channels (3 rows per partition instead of 4)", ir_ch=1)
    import irbaboon_amd as ib
    plan = ib.LargeFft(1 << 20, 1, device=dev)
    plan.forward(1)
    plan.inverse(1)
    plan.synchronize()
    plan.mark(0)
    for _ in range(50):
        plan.forward(1)
    plan.mark(1)
    f = plan.elapsed_ms() / 50
    plan.mark(0)
    for _ in range(50):
        plan.inverse(1)
    plan.mark(1)
    i = plan.elapsed_ms() / 50
    res["configs[4]_2^20_real_fft"] = {"forward_us": round(f * 1e3, 1), "inverse_us": round(i * 1e3, 1),
                                       "effective_GBps_min_traffic": round(8.0 * (1 << 20) / f / 1e6, 1),
                                       "note": "single 2^20-point real FFT (four-step, 3 launches); latency-bound"}
    plan.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--instances", type=int, default=2048, help="instances per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-single", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch                      # plumbing: device sync + torch.distributed (RCCL) for barrier/max
    from irbaboon_amd import FdlEngine, shard
    from irbaboon_amd.synth import decaying_ir, white_noise

    ranks = shard.Ranks()
    if ranks.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ranks.world}: for N > 1 launch one rank per GPU with "
                         f"python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    dev = ranks.local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)

    I, K, W, NB = args.instances, args.steps, args.warmup, 8
    eng = FdlEngine(I, 1, B, LH, max_blocks_per_call=NB, device=dev)
    info = eng.info
    lo = ranks.rank * I               # weak scaling: every GPU owns I instances of its own
    for i in range(I):
        eng.set_ir(i, decaying_ir(18000 + 2 * (lo + i), LH))
    d_in, d_out, stride = eng.device_io()
    x = np.stack([white_noise(2000 + 2 * (lo + i), NB * B) for i in range(I)])
    eng.upload(d_in, x)

    def step(k):
        off = (k % NB) * B * 4
        eng.process_device(d_in + off, d_out + off, stride, 1)

    for k in range(W):
        step(k)
    eng.synchronize()
    torch.cuda.synchronize()
    ranks.barrier()
    t0 = time.perf_counter()
    eng.mark(0)
    for k in range(K):
        step(W + k)
    eng.mark(1)
    eng.synchronize()
    torch.cuda.synchronize()
    ranks.barrier()
    wall = time.perf_counter() - t0
    ev_ms = eng.elapsed_ms()
    wall = ranks.max(wall)
    ev_ms_max = ranks.max(ev_ms)

    total_samples = float(I) * ranks.world * B * K
    value = total_samples / wall / 1e6
    launches_per_step = 1 if info.splits <= 1 else 3
    kern_ms = ev_ms_max / K           # fused path: one k_step_fused launch per step
    alg_bytes = I * info.algorithmic_bytes_per_block_step          # per launch: I * (8*P*N + 16*B)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            if t.get("instances") == I:
                traffic = t.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "Msamples/s (48 kHz, 256-blk, 2 s IR); achieved HBM GB/s vs roofline",
        "value": round(value, 2), "unit": "Msamples/s", "n_gpus": ranks.world, "steps": K, "warmup": W,
        "ms_per_step": round(wall / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[1] geometry (mono, B={B}, Lh={LH}, P={info.partitions}, N={info.fft_len}) "
                               f"x {I} independent instances per GPU, streaming: 1 block per instance per step",
                   "instances_per_gpu": I, "block": B, "ir_len": LH, "partitions": info.partitions,
                   "fft_len": info.fft_len, "parallelism": f"instance-shard x{ranks.world} (no collective)",
                   "launches_per_step": launches_per_step, "device_bytes_per_gpu": int(info.device_bytes)},
        "roofline": {"bound": "hbm", "kernel": "k_step_fused<512>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(kern_ms, 4)},
    }

    if ranks.rank == 0 and ranks.world == 1 and not args.no_single:
        # the literal configs[1]: ONE mono stream (cache-resident, launch/latency-bound)
        one = FdlEngine(1, 1, B, LH, max_blocks_per_call=NB, device=dev)
        one.set_ir(0, decaying_ir(18000, LH))
        di, do, st = one.device_io()
        one.upload(di, x[0])
        for k in range(50):
            one.process_device(di + (k % NB) * B * 4, do + (k % NB) * B * 4, st, 1)
        one.synchronize()
        n1 = 2000
        t1 = time.perf_counter()
        for k in range(n1):
            one.process_device(di + (k % NB) * B * 4, do + (k % NB) * B * 4, st, 1)
        one.synchronize()
        dt1 = time.perf_counter() - t1
        out["single_stream"] = {"value": round(n1 * B / dt1 / 1e6, 3), "unit": "Msamples/s",
                                "us_per_block": round(dt1 / n1 * 1e6, 2), "splits": one.info.splits,
                                "note": "one mono stream, one block per step, 1.5 MiB working set (cache-resident)"}
        one.close()

    if ranks.rank == 0 and ranks.world == 1 and not args.no_single:
        eng.close()
        out["other_configs"] = other_configs(dev, np, FdlEngine, decaying_ir, white_noise)

    if ranks.rank == 0 and ranks.world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        out["cpu_baseline_all_cores"] = cpu_baseline_multicore(out["cpu_baseline"], min(4.0, args.cpu_seconds))
    eng.close()
    ranks.close()
    if ranks.rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
