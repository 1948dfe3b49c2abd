"""Ad-hoc timing probes on the GPU box (not part of the bench contract).
usage: python tools/perf_probe.py offline|batched|single [args]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from irbaboon_amd import FdlEngine, _lib
from irbaboon_amd.synth import decaying_ir, white_noise


def timed(eng, d_in, d_out, stride, T, reps):
    eng.process_device(d_in, d_out, stride, T); eng.synchronize()
    eng.mark(0)
    for _ in range(reps):
        eng.process_device(d_in, d_out, stride, T)
    eng.mark(1)
    return eng.elapsed_ms() / reps


def offline(B=256, Lh=96000, T=3750, S=1, reps=5, same_data=0):
    """one call carrying T blocks per stream (time-tiled path); same_data=1: one IR and one signal for every stream (counter runs: the
    traffic is the same, the set-up of 2048 streams is not)"""
    eng = FdlEngine(S, 1, B, Lh, max_blocks_per_call=T)
    h0, x0 = decaying_ir(1, Lh), white_noise(5, T * B)
    for i in range(S):
        eng.set_ir(i, h0 if same_data else decaying_ir(1 + i, Lh))
    d_in, d_out, stride = eng.device_io()
    eng.upload(d_in, np.stack([x0 if same_data else white_noise(5 + i, T * B) for i in range(S)]))
    if reps > 50:
        timed(eng, d_in, d_out, stride, T, reps)        # sustained load first: the clock the chip settles at
    ms = timed(eng, d_in, d_out, stride, T, reps)
    P, N = eng.info.partitions, eng.info.fft_len
    flops = 8.0 * P * (N // 2) * T * S
    print(f"tiled B={B} Lh={Lh} P={P} T={T} S={S}: {ms:.3f} ms/call  {S*T*B/ms/1e3:.1f} Msamples/s  "
          f"{flops/ms/1e9:.2f} TFLOP/s (MAC only)  ring={eng.info.ring_slots}")


def single(B=256, Lh=96000, n=2000, flags=0):
    eng = FdlEngine(1, 1, B, Lh, max_blocks_per_call=1, flags=flags)
    eng.set_ir(0, decaying_ir(1, Lh))
    d_in, d_out, stride = eng.device_io()
    eng.upload(d_in, white_noise(5, B))
    eng.process_device(d_in, d_out, stride, 1); eng.synchronize()
    t0 = time.perf_counter()
    eng.mark(0)
    for _ in range(n):
        eng.process_device(d_in, d_out, stride, 1)
    eng.mark(1)
    ms = eng.elapsed_ms()
    wall = time.perf_counter() - t0
    print(f"single stream B={B} Lh={Lh} splits={eng.info.splits}: device {ms/n*1e3:.2f} us/block, host-inclusive {wall/n*1e6:.2f} us/block")


def few(I=1, C=2, B=128, Lh=480000, n=2000, flags=0):
    """few streams, one block per call, device-resident: the latency path (configs[2] by default)"""
    eng = FdlEngine(I, C, B, Lh, max_blocks_per_call=1, flags=flags)
    for i in range(I):
        eng.set_ir(i, np.stack([decaying_ir(1 + 2 * i + c, Lh) for c in range(C)]))
    d_in, d_out, stride = eng.device_io()
    eng.upload(d_in, np.stack([white_noise(5 + s, B) for s in range(I * C)]))
    for _ in range(20):
        eng.process_device(d_in, d_out, stride, 1)
    eng.synchronize()
    eng.mark(0)
    for _ in range(n):
        eng.process_device(d_in, d_out, stride, 1)
    eng.mark(1)
    ms = eng.elapsed_ms()
    alg = I * C * eng.info.algorithmic_bytes_per_block_step
    env = {k: v for k, v in os.environ.items() if k.startswith("IRB_")}
    print(f"few I={I} C={C} B={B} Lh={Lh} P={eng.info.partitions} wgs/stream={eng.info.splits} flags={flags} {env}: "
          f"{ms/n*1e3:.2f} us/step  {I*C*B/(ms/n)/1e3:.1f} Msamples/s  {alg/(ms/n)/1e6:.0f} GB/s algorithmic")


def twoqueue(I=2048, B=256, Lh=96000, n=300, parts=2):
    """the headline batch as `parts` engines of I/parts instances, each on its own HIP stream, stepped in turn with no
    join between them: in steady state one part's launch gap / ramp-up / tail overlaps the other's streaming.  Shows how
    much of the single-launch step time is per-launch fixed cost (the roofline line itself stays single-launch)."""
    per = I // parts
    engs = []
    for p in range(parts):
        e = FdlEngine(per, 1, B, Lh, max_blocks_per_call=1)
        for i in range(per):
            e.set_ir(i, decaying_ir(1 + p * per + i, Lh))
        d_in, d_out, stride = e.device_io()
        e.upload(d_in, np.stack([white_noise(5 + p * per + s, B) for s in range(per)]))
        engs.append((e, d_in, d_out, stride))
    def steps(k):
        for _ in range(k):
            for e, di, do, st in engs:
                e.process_device(di, do, st, 1)
    steps(20)
    for e, *_ in engs:
        e.synchronize()
    t0 = time.perf_counter()
    steps(n)
    for e, *_ in engs:
        e.synchronize()
    dt = (time.perf_counter() - t0) / n
    alg = I * engs[0][0].info.algorithmic_bytes_per_block_step
    print(f"twoqueue I={I} parts={parts} B={B} Lh={Lh}: {dt*1e6:.1f} us/step (wall)  {I*B/dt/1e6:.1f} Msamples/s  {alg/dt/1e9:.0f} GB/s algorithmic")


def stereo(I=1024, B=256, Lh=96000, flags=0):
    """stereo instances with a mono IR (the plug-in's real-time layout), one block per call"""
    eng = FdlEngine(I, 2, B, Lh, max_blocks_per_call=1, flags=flags)
    for i in range(I):
        eng.set_ir(i, decaying_ir(1 + i, Lh))
    d_in, d_out, stride = eng.device_io()
    eng.upload(d_in, np.stack([white_noise(5 + s, B) for s in range(2 * I)]))
    ms = timed(eng, d_in, d_out, stride, 1, 100)
    print(f"stereo/mono-IR I={I} B={B} Lh={Lh} flags={flags}: {ms*1e3:.1f} us/step  {2*I*B/ms/1e3:.1f} Msamples/s")


def rt(host_block=256, n=2000):
    """processBlock-shaped adapter: wall time per host callback (stereo, 8 partitions, rolling IR refresh)"""
    from irbaboon_amd import RtConvolver
    r = RtConvolver(2, 256, 2048)
    r.prepareToPlay(host_block)
    buf = np.stack([white_noise(1, host_block), white_noise(2, host_block)])
    for _ in range(50):
        b = buf.copy(); r.processBlock(b)
    ts = []
    for _ in range(n):
        b = buf.copy()
        t0 = time.perf_counter(); r.processBlock(b); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print(f"rt adapter hostBlock={host_block}: median {np.median(ts):.1f} us, p99 {np.percentile(ts, 99):.1f} us, max {ts.max():.1f} us per callback "
          f"(deadline {host_block/48000*1e6:.0f} us at 48 kHz)")


def rtc(host_block=256, n=3000, ir_ring_len=2048, channels=2):
    """the same callbacks timed in C (irb_rt_time_callbacks): what bench.py's rt_processBlock entry reports"""
    from irbaboon_amd import RtConvolver
    r = RtConvolver(channels, 256, ir_ring_len)
    r.prepareToPlay(host_block)
    r.time_callbacks(host_block, 200)
    c0 = r.counters()
    ts = r.time_callbacks(host_block, n)
    c1 = r.counters()
    cb = c1[0] - c0[0]
    env = {k: v for k, v in os.environ.items() if k.startswith("IRB_")}
    print(f"rt adapter (C loop) hostBlock={host_block} ring={ir_ring_len} ch={channels} {env}: median {np.median(ts):.1f} us, p99 {np.percentile(ts, 99):.1f} us, max {ts.max():.1f} us; "
          f"per callback: {(c1[1]-c0[1])/cb:.2f} launches, {(c1[2]-c0[2])/cb:.2f} waits, {(c1[3]-c0[3])/cb:.2f} copies")


def hostapi(I=2048, B=256, Lh=96000, T=1, reps=30):
    """PCIe-inclusive rate through the host-pointer API (irb_fdl_process): pack, H2D, kernels, D2H, unpack"""
    import ctypes as C
    from irbaboon_amd._lib import lib, planar_ptrs, check
    eng = FdlEngine(I, 1, B, Lh, max_blocks_per_call=T)
    h0 = decaying_ir(1, Lh)
    for i in range(I):
        eng.set_ir(i, h0)
    x = np.stack([white_noise(5 + s, T * B) for s in range(I)])
    y = np.empty_like(x)
    xp, yp = planar_ptrs(x), planar_ptrs(y)
    check(lib().irb_fdl_process(eng._h, xp, yp, T))
    t0 = time.perf_counter()
    for _ in range(reps):
        check(lib().irb_fdl_process(eng._h, xp, yp, T))
    dt = (time.perf_counter() - t0) / reps
    from irbaboon_amd.engine import pinned_empty
    xp2, yp2 = pinned_empty(x.shape), pinned_empty(x.shape)
    xp2[:] = x
    eng.process_packed(xp2, yp2)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.process_packed(xp2, yp2)
    dp = (time.perf_counter() - t0) / reps
    eng.process_packed(x, y)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.process_packed(x, y)
    dq = (time.perf_counter() - t0) / reps
    print(f"   packed API: pinned {dp*1e3:.3f} ms ({I*T*B/dp/1e6:.1f} Msamples/s), pageable {dq*1e3:.3f} ms ({I*T*B/dq/1e6:.1f} Msamples/s)")
    print(f"host API I={I} B={B} T={T}: {dt*1e3:.3f} ms per call  {I*T*B/dt/1e6:.1f} Msamples/s PCIe-inclusive "
          f"({2*I*T*B*4/dt/1e9:.1f} GB/s over PCIe both ways)")


def lfft(logn=20, batch=1, reps=50):
    """device-resident large real FFT: forward and inverse, and the full one-shot deconvolve vs the oracle"""
    import irbaboon_amd as ib
    N = 1 << logn
    plan = ib.LargeFft(N, batch)
    plan.forward(batch); plan.inverse(batch); plan.synchronize()
    plan.mark(0)
    for _ in range(reps):
        plan.forward(batch)
    plan.mark(1)
    f = plan.elapsed_ms() / reps
    plan.mark(0)
    for _ in range(reps):
        plan.inverse(batch)
    plan.mark(1)
    i = plan.elapsed_ms() / reps
    # minimum traffic: read N floats + write N+2 floats (forward)
    print(f"lfft N=2^{logn} batch={batch}: forward {f*1e3:.1f} us, inverse {i*1e3:.1f} us, "
          f"effective {batch*8.0*N/f/1e6:.1f} GB/s (forward, min traffic 8N bytes)")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle
    from irbaboon_amd.synth import exp_sine_sweep
    sweep = exp_sine_sweep(3 * N // 4, 48000, 20.0, 24000, fade_from=22000, n_total=N)
    cap = np.roll(sweep, 100) * 0.5
    for smoothing in (False, True):
        t0 = time.perf_counter(); ib.deconvolve(cap, sweep, 48000, smoothing); g = time.perf_counter() - t0
        t0 = time.perf_counter(); oracle.deconvolve(cap, sweep, 48000, smoothing); c = time.perf_counter() - t0
        print(f"deconvolve N=2^{logn} smoothing={smoothing}: GPU path {g*1e3:.1f} ms (incl. plan + PCIe + host smoothing), oracle CPU {c*1e3:.1f} ms")


if __name__ == "__main__":
    mode = sys.argv[1]
    args = [int(a) for a in sys.argv[2:]]
    {"offline": offline, "single": single, "lfft": lfft, "stereo": stereo, "rt": rt, "rtc": rtc, "hostapi": hostapi, "few": few, "twoqueue": twoqueue}[mode](*args)
