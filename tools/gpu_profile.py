#!/usr/bin/env python
"""Runs on the GPU box (via gpurun): rocprofv3 evidence for one bench.py workload, summarised.

    python tools/gpu_profile.py <tag> [--pmc-steps 10] [-- bench args ...]

 1. `rocprofv3 --kernel-trace --stats` around `python bench.py --no-cpu-baseline <bench args>` with bench.py's
    DEFAULT --steps / --warmup / ramp (the same command the driver times), so that the average kernel duration of the
    trace can be compared with the driver's line;
 2. separate `--kernel-trace --pmc` passes (never combined with other trace domains) on a short run for FETCH_SIZE,
    WRITE_SIZE, L2 hit/miss and the SQ wait / LDS counters;
 3. `gpurun_out/prof_<tag>/summary.json`: per hot kernel the average duration, HBM traffic per launch
    (FETCH_SIZE KiB x 1024 x 2 -- gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md -- + WRITE_SIZE KiB x
    1024), traffic / algorithmic bytes, L2 hit rate, and the bench line of the profiled run.
Copy the summaries to profiles/ to have them judged.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
from pathlib import Path

R = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parents[1]))
HOT = ("starlin", "march", "star25d", "vecpt", "naive", "twostep")


def run(cmd, log):
    env = dict(os.environ, TMPDIR="/tmp")
    with open(log, "w") as f:
        return subprocess.run(cmd, cwd="/tmp", env=env, stdout=f, stderr=subprocess.STDOUT, timeout=900).returncode


def main():
    argv = sys.argv[1:]
    tag = argv.pop(0)
    pmc_steps = 10
    if argv and argv[0] == "--pmc-steps":
        pmc_steps = int(argv[1]); argv = argv[2:]
    if argv and argv[0] == "--":
        argv = argv[1:]
    out = R / "gpurun_out" / f"prof_{tag}"
    out.mkdir(parents=True, exist_ok=True)
    bench = [sys.executable, str(R / "bench.py"), "--no-cpu-baseline"] + argv
    run(["rocprofv3", "--kernel-trace", "--stats", "-f", "csv", "-d", str(out / "stats"), "--"] + bench, out / "stats.log")
    short = bench + ["--steps", str(pmc_steps), "--warmup", "2", "--ramp-secs", "0", "--no-probe"]
    passes = ["FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum TCC_MISS_sum", "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum",
              "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE",
              "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES"]
    for p in passes:
        name = p.replace(" ", "_")[:40]
        run(["rocprofv3", "--kernel-trace", "--pmc", *p.split(), "-f", "csv", "-d", str(out / f"pmc_{name}"), "--"] + short,
            out / f"pmc_{name}.log")
    # ---- summarise
    pmc = {}
    for f in glob.glob(str(out / "pmc_*" / "**" / "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            for c, v in d.items():
                pmc.setdefault(k, {})[c] = {"avg": sum(v) / len(v), "n": len(v)}
    json.dump({k[:80]: v for k, v in pmc.items()}, open(out / "pmc_summary.json", "w"), indent=1)
    stats = {}
    for f in glob.glob(str(out / "stats" / "**" / "*kernel_stats.csv"), recursive=True):
        (out / "kernel_stats.csv").write_text(open(f).read())
        for r in csv.DictReader(open(f)):
            stats[r["Name"]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) * 1e-6, "min_ms": float(r["MinNs"]) * 1e-6,
                                "max_ms": float(r["MaxNs"]) * 1e-6, "pct": float(r["Percentage"])}
    line = None
    for l in open(out / "stats.log"):
        if l.startswith("{") and '"metric"' in l:
            line = json.loads(l)
    if line is not None:
        json.dump(line, open(out / "bench_line.json", "w"), indent=1)
    summ = {"tag": tag, "bench_args": argv, "kernels": {},
            "correction": "FETCH_SIZE (KiB) x1024 x2 (gfx950 reports half of wide coalesced reads); WRITE_SIZE (KiB) x1024",
            "bench_line_of_the_profiled_run": {k: line.get(k) for k in ("value", "ms_per_step", "steps", "warmup", "roofline", "step_ms", "config")} if line else None}
    alg = line["roofline"]["algorithmic_bytes_per_launch"] if line else None
    tot_ms, tot_traffic = 0.0, 0.0
    for k, s in stats.items():
        if not any(h in k for h in HOT) or s["calls"] < 3:
            continue
        d = pmc.get(k, {})
        e = {"calls": s["calls"], "avg_ms": round(s["avg_ms"], 4), "min_ms": round(s["min_ms"], 4), "max_ms": round(s["max_ms"], 4)}
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            e["fetch_bytes_per_launch_corrected"] = d["FETCH_SIZE"]["avg"] * 1024 * 2
            e["write_bytes_per_launch"] = d["WRITE_SIZE"]["avg"] * 1024
            e["traffic_bytes_per_launch"] = e["fetch_bytes_per_launch_corrected"] + e["write_bytes_per_launch"]
            tot_traffic += e["traffic_bytes_per_launch"]
        if "TCC_HIT_sum" in d:
            e["l2_hit_rate"] = round(d["TCC_HIT_sum"]["avg"] / max(1.0, d["TCC_HIT_sum"]["avg"] + d["TCC_MISS_sum"]["avg"]), 4)
        if "SQ_WAIT_ANY" in d and d.get("SQ_WAVE_CYCLES", {}).get("avg"):
            e["wait_any_frac"] = round(d["SQ_WAIT_ANY"]["avg"] / d["SQ_WAVE_CYCLES"]["avg"], 4)
        for c in ("SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum",
                  "TCC_EA0_WRREQ_64B_sum"):
            if c in d:
                e[c.lower()] = d[c]["avg"]
        tot_ms += s["avg_ms"]
        summ["kernels"][k[:120]] = e
    if alg:
        summ["algorithmic_bytes_per_step"] = alg
        summ["sum_of_hot_kernel_avg_ms"] = round(tot_ms, 4)
        summ["algorithmic_GBps_at_rocprof_duration"] = round(alg / (tot_ms * 1e-3) * 1e-9, 1) if tot_ms else None
        summ["roofline_frac_at_rocprof_duration"] = round(alg / (tot_ms * 1e-3) * 1e-9 / 8000.0, 4) if tot_ms else None
        if tot_traffic:
            summ["traffic_bytes_per_step"] = tot_traffic
            summ["traffic_over_algorithmic"] = round(tot_traffic / alg, 4)
    json.dump(summ, open(out / "summary.json", "w"), indent=1)
    print(json.dumps(summ, indent=1)[:3000])


if __name__ == "__main__":
    main()
