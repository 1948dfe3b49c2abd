"""Turns the raw output of tools/profile_r03.sh (gpurun_out/prof_r03/...) into the small round-3 summaries committed
under profiles/.  HBM traffic follows MI355X_MICROARCH.md section HBM: FETCH_SIZE and
WRITE_SIZE from separate --pmc passes, values in KiB, FETCH_SIZE doubled on gfx950 for wide coalesced streaming
reads (16 B per lane), WRITE_SIZE exact.
gpurun only ADDS files to the local gpurun_out/: remove gpurun_out/prof_r03 and gpurun_out/tiled_prof here before the profiling
call, or files of earlier calls are mixed in.
usage: python profiles/collect_r03.py [gpurun_out]"""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def stats_csv(src_dir):
    """the kernel_stats file of the profiled program itself: child processes (compiled examples, the bench's PMC
    children) leave small files of their own next to it"""
    f = glob.glob(os.path.join(src_dir, "*", "*_kernel_stats.csv"))
    if not f:
        return None
    newest = max(os.path.getmtime(x) for x in f)               # files of one call are written within seconds of each other
    return max((x for x in f if newest - os.path.getmtime(x) < 120), key=os.path.getsize)


def counters(src_dir, kernel):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src_dir, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {"dispatches": len(v), "mean": statistics.mean(v), "min": min(v), "max": max(v)} for k, v in acc.items()}


def kernel_names(path):
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(path))}


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(HERE), "gpurun_out")
    P = os.path.join(out_dir, "prof_r03")
    shutil.copy(os.path.join(P, "bench.json"), os.path.join(HERE, "r03_bench.json"))
    shutil.copy(os.path.join(P, "bench_under_trace.json"), os.path.join(HERE, "r03_bench_under_trace.json"))
    for src, dst in (("bench_kt", "r03_bench_default_kernel_stats.csv"), ("tests_kt", "r03_gputest_kernel_stats.csv"),
                     ("rt256_kt", "r03_rt_callback_256_kernel_stats.csv"), ("rt512_kt", "r03_rt_callback_512_kernel_stats.csv"),
                     ("lfft_kt", "r03_lfft_kernel_stats.csv")):
        f = stats_csv(os.path.join(P, src))
        if f:
            shutil.copy(f, os.path.join(HERE, dst))
    # which bench kernels does the GPU test suite also run?  (only our kernels: namespace irb)
    bench = kernel_names(os.path.join(HERE, "r03_bench_default_kernel_stats.csv"))
    tests = kernel_names(os.path.join(HERE, "r03_gputest_kernel_stats.csv"))
    lines = ["kernel of the default `python bench.py` run | bench calls | bench avg us | launches in `pytest -m gpu`"]
    missing = 0
    for name, (calls, avg) in sorted(bench.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        if "irb::" not in name:
            continue
        t = tests.get(name)
        missing += t is None
        lines.append(f"{name} | {calls} | {avg / 1e3:.2f} | {t[0] if t else 'NOT RUN BY THE TESTS'}")
    lines.append(f"bench kernels without a parity-test launch: {missing}")
    open(os.path.join(HERE, "r03_kernel_coverage.txt"), "w").write("\n".join(lines) + "\n")
    # headline kernel: HBM traffic per launch
    k = "k_step_fused"
    c = counters(os.path.join(P, "pmc_FETCH_SIZE"), k)
    c.update(counters(os.path.join(P, "pmc_WRITE_SIZE"), k))
    fetch_b = c["FETCH_SIZE"]["mean"] * 1024 * 2
    write_b = c["WRITE_SIZE"]["mean"] * 1024
    avg_ns = [v[1] for n, v in bench.items() if "k_step_fused<512, 2, true>" in n]
    calls = [v[0] for n, v in bench.items() if "k_step_fused<512, 2, true>" in n]
    alg = 2048 * (8 * 375 * 512 + 16 * 256)
    summary = {"kernel": "k_step_fused<512,2,true>", "instances": 2048, "kernel_trace": {"calls": calls[0], "average_ns": avg_ns[0]},
               "FETCH_SIZE": c["FETCH_SIZE"], "WRITE_SIZE": c["WRITE_SIZE"],
               "read_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b, "hbm_bytes_per_launch": fetch_b + write_b,
               "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (fetch_b + write_b) / alg,
               "achieved_GBps_from_kernel_trace": alg / avg_ns[0]}
    json.dump(summary, open(os.path.join(HERE, "r03_pmc_summary.json"), "w"), indent=1)
    json.dump({"instances": 2048, "hbm_bytes_per_launch": fetch_b + write_b, "source": "profiles/r03_pmc_summary.json"},
              open(os.path.join(HERE, "hbm_traffic.json"), "w"))
    # time-tiled MAC: counters of the packed-FMA kernel and of round 2's
    tiled = {}
    for f in sorted(glob.glob(os.path.join(out_dir, "tiled_prof", "pmc_*_k*_*"))):
        if not os.path.isdir(f):
            continue
        tag = os.path.basename(f)[4:].rsplit("_", 1)[0]
        cc = counters(f, "k_cmac_tiled")
        tiled.setdefault(tag, {}).update({k: v["mean"] for k, v in cc.items()})
    for tag, m in tiled.items():
        if "SQ_WAVE_CYCLES" in m:
            m["derived"] = {"share_issuing": m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], "share_waiting_on_memory_or_barrier": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
                            "share_issue_stalled": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
                            "valu_share_of_issue": m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_ACTIVE_INST_ANY"],
                            "l2_hit_rate": m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) if "TCC_HIT_sum" in m else None}
    json.dump({"note": "rocprofv3 --pmc passes over tools/perf_probe.py offline <B Lh T S> (tools/tiled_profile.sh); keys: <B>_<Lh>_<T>_<S>_k<IRB_TILED_KERNEL> "
                       "(k1 = k_cmac_tiled_pk, k0 = round 2's k_cmac_tiled); SQ_* in quad-cycles summed over waves, per launch",
               "counters_mean_per_launch": tiled}, open(os.path.join(HERE, "r03_tiled_mac_pmc.json"), "w"), indent=1)
    for src, dst in (("tiled_pmc_cfg2.txt", "r03_tiled_kernels_cfg2.txt"), ("tiled_pmc_cfg3.txt", "r03_tiled_kernels_cfg3.txt"), ("tiled_stamps.txt", "r03_tiled_stamps.txt"),
                     ("valu_peak.txt", "r03_valu_peak.txt"), ("rt_ab.txt", "r03_rt_callback_ab.txt"), ("handoff_under_load.txt", "r03_handoff_under_load.txt"),
                     ("mfma_trace.txt", "r03_mfma_kernel_trace.txt"), ("mfma_stamps.txt", "r03_mfma_stamps.txt"), ("mfma_pmc.txt", "r03_mfma_pmc.txt"),
                     ("mfma_peak.txt", "r03_mfma_peak.txt"), ("mfma_ab.txt", "r03_mfma_ab.txt")):
        f = os.path.join(P, src)
        if os.path.exists(f):
            shutil.copy(f, os.path.join(HERE, dst))
    print(open(os.path.join(HERE, "r03_kernel_coverage.txt")).read())
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
