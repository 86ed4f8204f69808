#!/usr/bin/env python
"""GPU box: where does ONE kernel shape of a generic-registry solution spend its cycles?  Runs the shape a few times (time_part) under
separate `rocprofv3 --kernel-trace --pmc` passes (never combined with another trace domain) and prints / writes per-kernel averages of
the SQ counters: busy / wave cycles, waiting vs issuing, VALU / LDS / VMEM active cycles, instruction counts, LDS conflicts.
    python tools/variant_pmc.py --stencil cube --variant box_v4_z128_y16_r1_nt_w2 [--size 512] [--out gpurun_out/pmc_cube]"""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys
from pathlib import Path

R = Path(__file__).resolve().parents[1]
PASSES = ["SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY",
          "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA",
          "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS",
          "SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32",
          "GRBM_GUI_ACTIVE FETCH_SIZE"]


def child(args):
    sys.path.insert(0, str(R))
    from yask_amd import yk_factory
    fac = yk_factory(args.stencil)
    s = fac.new_solution(fac.new_env())
    s.set_overall_domain_size_vec([args.size] * 3)
    s.apply_command_line_options("-no-auto_tune")
    s.prepare_solution()
    for k, v in enumerate(s.get_vars()):
        v.set_elements_hash(1.0 + 0.25 * k, 0.1, hash_id=k)
    names = s.get_kernel_variant_names(args.part)
    vi = names.index(args.variant)
    ms = s.time_part(args.part, vi, 0, 0, args.reps)
    print(json.dumps({"variant": args.variant, "ms": ms}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stencil", required=True)
    ap.add_argument("--variant", required=True)
    ap.add_argument("--part", type=int, default=0)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    out = Path(args.out or R / "gpurun_out" / f"pmc_{args.stencil}_{args.variant}")
    out.mkdir(parents=True, exist_ok=True)
    cmd = [sys.executable, str(Path(__file__).resolve()), "--child", "--stencil", args.stencil, "--variant", args.variant, "--part", str(args.part),
           "--size", str(args.size), "--reps", str(args.reps)]
    env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=str(R))
    res = {}
    for i, p in enumerate(PASSES):
        d = out / f"pass{i}"
        with open(out / f"pass{i}.log", "w") as f:
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *p.split(), "-f", "csv", "-d", str(d), "--"] + cmd, cwd="/tmp", env=env, stdout=f,
                           stderr=subprocess.STDOUT, timeout=600)
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(str(d / "**" / "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, dd in acc.items():
            if "_kernel<" not in k or "bw_probe" in k:
                continue
            for c, v in dd.items():
                res.setdefault(k[:90], {})[c] = round(sum(v) / len(v), 1)
        dur = collections.defaultdict(list)
        for f in glob.glob(str(d / "**" / "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                dur[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
        for k, v in dur.items():
            if k in res:
                res[k][f"avg_ms_pass{i}"] = round(sum(v) / len(v), 4)
        subprocess.run(["rm", "-rf", str(d)])
    json.dump(res, open(out / "summary.json", "w"), indent=1)
    for k, v in res.items():
        print(k)
        print("   ", json.dumps(v))
        if "SQ_BUSY_CYCLES" in v and "SQ_ACTIVE_INST_VALU" in v:
            # SQ_* cycle counters are summed over SEs / SIMDs differently by revision: report ratios that do not depend on it
            wc = v.get("SQ_WAVE_CYCLES", 0)
            print("    of wave-cycles: waiting %.2f, issue-stalled %.2f, issuing %.2f | active cycles VALU : LDS : VMEM = %.3g : %.3g : %.3g (vs busy %.3g)" % (
                v.get("SQ_WAIT_ANY", 0) / max(wc, 1), v.get("SQ_WAIT_INST_ANY", 0) / max(wc, 1), v.get("SQ_ACTIVE_INST_ANY", 0) / max(wc, 1),
                v["SQ_ACTIVE_INST_VALU"], v.get("SQ_ACTIVE_INST_LDS", 0), v.get("SQ_ACTIVE_INST_VMEM", 0), v["SQ_BUSY_CYCLES"]))


if __name__ == "__main__":
    main()
