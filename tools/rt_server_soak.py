"""Soak of the resident real-time server (k_rt_server): tens of thousands of callbacks with random pauses around its idle
timeout (so that generations keep ending and starting under the audio thread's feet), random host block sizes per run, the IR
replaced at random times, two adapters alive at once -- every sample compared with the one-launch-per-callback form.
usage: python tools/rt_server_soak.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from irbaboon_amd import RtConvolver
from irbaboon_amd.synth import decaying_ir, white_noise

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
os.environ["IRB_RT_SERVER_IDLE_US"] = "300"
t0, runs, callbacks, launches = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    hb = int(rng.choice([32, 64, 100, 256, 300, 512, 1024]))
    n_cb = int(rng.integers(200, 1500))
    x = np.stack([white_noise(int(rng.integers(1 << 30)), n_cb * hb) for _ in range(2)])
    events = {int(k): decaying_ir(int(rng.integers(1 << 30)), 2048) for k in rng.integers(0, n_cb, 4)}
    pauses = {int(k): float(rng.choice([0.0001, 0.0004, 0.002])) for k in rng.integers(0, n_cb, 40)}
    outs = []
    for resident in ("1", "0"):
        os.environ["IRB_RT_RESIDENT"] = resident
        a, b = RtConvolver(2, 256, 2048), RtConvolver(2, 256, 2048)          # two plug-in instances, interleaved callbacks
        a.prepareToPlay(hb); b.prepareToPlay(hb)
        ya, yb = np.empty_like(x), np.empty_like(x)
        for k in range(n_cb):
            if k in events:
                a.setIR(events[k]); b.setIR(events[k][::-1].copy())
            if resident == "1" and k in pauses:
                time.sleep(pauses[k])
            u = np.ascontiguousarray(x[:, k * hb:(k + 1) * hb]); v = u[::-1].copy()
            a.processBlock(u); b.processBlock(v)
            ya[:, k * hb:(k + 1) * hb], yb[:, k * hb:(k + 1) * hb] = u, v
        if resident == "1":
            launches += a.counters()[1] + b.counters()[1]
            callbacks += 2 * n_cb
        a.close(); b.close()
        outs.append((ya, yb))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), (hb, n_cb)
    runs += 1
print(f"{runs} runs, {callbacks} resident callbacks on two adapters at once, {launches} server generations started "
      f"(idle timeout 300 us, pauses up to 2 ms): every sample equal to the one-launch form; {time.time()-t0:.0f} s")
