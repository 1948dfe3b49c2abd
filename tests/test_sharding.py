"""Multi-rank host logic on CPU: gloo, world_size 2.  Instances shard with no data-path
collective; ranks only meet at the barrier and the max-over-ranks of their timings."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from irbaboon_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,world", [(1024, 8), (1024, 3), (5, 8), (7, 2), (0, 4)])
def test_instance_range_is_a_partition(n, world):
    seen = []
    for r in range(world):
        lo, hi = shard.instance_range(n, r, world)
        assert 0 <= lo <= hi <= n
        seen += list(range(lo, hi))
    assert seen == list(range(n))
    sizes = [shard.instance_range(n, r, world)[1] - shard.instance_range(n, r, world)[0] for r in range(world)]
    assert max(sizes) - min(sizes) <= 1
    for i in range(n):
        lo, hi = shard.instance_range(n, shard.owner_of(i, n, world), world)
        assert lo <= i < hi


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np
    from irbaboon_amd import shard
    from irbaboon_amd.synth import decaying_ir, white_noise
    import oracle
    ranks = shard.Ranks(backend="gloo")
    lo, hi = shard.instance_range(6, ranks.rank, ranks.world)
    sums = {{}}
    for g in range(lo, hi):                       # each rank convolves only the instances it owns
        x = white_noise(1000 + 2 * g, 2048)[None]; h = decaying_ir(9000 + 2 * g, 700)[None]
        y, _ = oracle.convolve_periodic(x, h, 128)
        sums[g] = float(np.sum(y.astype(np.float64) ** 2))
    ranks.barrier()
    t = ranks.max(1.0 + ranks.rank)               # slowest rank sets the step time
    n = ranks.sum(hi - lo)
    json.dump({{"rank": ranks.rank, "range": [lo, hi], "sums": sums, "tmax": t, "n": n}}, open(sys.argv[1] + str(ranks.rank), "w"))
    ranks.close()
""")


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script), str(tmp_path / "out")]
    subprocess.run(cmd, check=True, env=env, timeout=300, capture_output=True)
    import json
    import oracle
    from irbaboon_amd.synth import decaying_ir, white_noise
    res = [json.load(open(str(tmp_path / "out") + str(r))) for r in range(2)]
    assert res[0]["range"] == [0, 3] and res[1]["range"] == [3, 6]
    assert res[0]["tmax"] == res[1]["tmax"] == 2.0 and res[0]["n"] == 6.0
    merged = {}
    for r in res:
        merged.update(r["sums"])
    for g in range(6):                             # sharded results == single-process results
        x = white_noise(1000 + 2 * g, 2048)[None]
        h = decaying_ir(9000 + 2 * g, 700)[None]
        y, _ = oracle.convolve_periodic(x, h, 128)
        assert merged[str(g)] == pytest.approx(float(np.sum(y.astype(np.float64) ** 2)), rel=1e-12)
