#!/usr/bin/env python
"""Performance / validation harness for the cdna4_hip kernel libraries -- the counterpart of the reference's
src/kernel/yask_main.cpp (which cannot be relinked against a foreign library: it reaches into StencilContext
internals) with the same option names, the same trial protocol and the same log keys, so that the
reference's tooling keeps working on its output (src/kernel/yask.sh:595-613 greps `best-throughput`,
`mid-throughput`, `TEST PASSED|FAILED`, `YASK DONE`; utils/lib/YaskUtils.pm:36-110 parses the key: value lines).

    python -m yask_amd.harness -stencil iso3dfd -g 512 -trial_steps 50 -num_trials 3 [-validate] [yask options]

Harness options (yask_main.cpp:70-149): -help -stencil -num_trials|-t -trial_steps|-dt -trial_time -warmup
-no-warmup -pre_auto_tune -no-pre_auto_tune -init_seed -validate -sleep.  Everything else goes to
yk_solution::apply_command_line_options (-g/-l/-nr/-b..., -overlap_comms, -hip_variant, ...); leftovers are an
error, as in yask_main.cpp:214-217.  -validate re-runs the same steps with the generic one-point-per-thread
kernel on a second solution (the role of the reference's scalar `run_ref`, yask_main.cpp:562-644) and compares
with the reference's rule (compare_data, epsilon 1e-3)."""
from __future__ import annotations

import math
import sys
import time

DIV = "─" * 60 + "\n"


def num_str(x):
    """Engineering suffixes like make_num_str (src/common/common_utils.cpp:104-150)."""
    if isinstance(x, int) and -1000 < x < 1000:
        return str(x)
    x = float(x)
    if x == 0:
        return "0"
    a = abs(x)
    for lim, suf in ((1e18, "E"), (1e15, "P"), (1e12, "T"), (1e9, "G"), (1e6, "M"), (1e3, "K")):
        if a >= lim:
            return f"{x / lim:g}{suf}"
    if a >= 1:
        return f"{x:g}"
    for lim, suf in ((1e-3, "m"), (1e-6, "u"), (1e-9, "n")):
        if a >= lim:
            return f"{x / lim:g}{suf}"
    return f"{x:g}"


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    o = {"stencil": "iso3dfd", "num_trials": 3, "trial_steps": 10, "trial_time": 0.0, "warmup": True,
         "pre_auto_tune": False, "init_seed": 0.1, "validate": False, "sleep": 0}
    rest = []
    i = 0
    while i < len(argv):
        a = argv[i]
        def val():
            nonlocal i
            i += 1
            if i >= len(argv):
                raise SystemExit(f"YASK Kernel: YASK error: option '{a}' requires a value.")
            return argv[i]
        if a in ("-help", "-h", "--help"):
            print(__doc__)
            return 0
        elif a == "-stencil": o["stencil"] = val()
        elif a in ("-num_trials", "-t"): o["num_trials"] = int(val())
        elif a in ("-trial_steps", "-dt"): o["trial_steps"] = int(val())
        elif a == "-trial_time": o["trial_time"] = float(val())
        elif a == "-init_seed": o["init_seed"] = float(val())
        elif a == "-sleep": o["sleep"] = int(val())
        elif a in ("-validate", "-v"): o["validate"] = True
        elif a == "-warmup": o["warmup"] = True
        elif a == "-no-warmup": o["warmup"] = False
        elif a == "-pre_auto_tune": o["pre_auto_tune"] = True
        elif a == "-no-pre_auto_tune": o["pre_auto_tune"] = False
        else: rest.append(a)
        i += 1

    from . import yk_factory, dist as ydist
    rank, _, world = ydist.init_process_group()
    out = sys.stdout if rank == world - 1 else open("/dev/null", "w")     # output from the last rank, like the reference
    try:
        fac = yk_factory(o["stencil"])
        env, transport = ydist.new_env(fac, "rccl")
        soln = fac.new_solution(env)
        if world == 1:      # (the var-placement search is opt-in in the library; see yask_main_hip.cpp)
            soln.apply_command_line_options("-hip_placement_trials 6")
        rem = soln.apply_command_line_options(" ".join(rest))
        if rem:
            raise RuntimeError(f"YASK error: extraneous parameter(s): '{rem}'; run with '-help' option for usage")
        out.write(DIV + f"YASK – Yet Another Stencil Kit, kernel library {fac.get_version_string()}\n"
                  f"Stencil name: {soln.get_name()}\nTarget: {soln.get_target()}\nElement size: {soln.get_element_bytes()} bytes\n"
                  f"Num ranks: {env.get_num_ranks()}; halo transport: {transport}\n")
        soln.prepare_solution()
        out.write(f"Kernel variant(s): " + ", ".join(soln.get_kernel_variant(p) for p in range(soln.get_num_parts())) + "\n")

        def init_vars(s):
            # the role of init_vars()/set_all_elements_in_seq(-init_seed) (setup.cpp:1023-1040), but independent of
            # the storage layout: var k = 1 + k/4 + seed*hash(logical index), the same data in both solutions
            # when validating; O(1) values with a small perturbation keep coefficient vars away from zero
            for k, v in enumerate(s.get_vars()):
                v.set_elements_hash(1.0 + 0.25 * k, o["init_seed"], hash_id=k)

        init_vars(soln)
        if o["pre_auto_tune"]:
            out.write(DIV + "Running the auto-tuner over the compiled tile shapes...\n")
            soln.run_auto_tuner_now(False)
            out.write("Kernel variant(s) after tuning: " + ", ".join(soln.get_kernel_variant(p) for p in range(soln.get_num_parts())) + "\n")
        steps = o["trial_steps"]
        t0 = 0
        if o["warmup"]:
            out.write(DIV + "Running warmup step(s)...\n")
            tw = time.perf_counter()
            soln.run_solution(t0, t0)
            t0 += 1
            if o["trial_time"] > 0:          # calibrate the number of steps from the warm-up rate
                soln.run_solution(t0, t0 + 4)
                t0 += 5
                st = soln.get_stats()
                rate = 5 / max(st.get_elapsed_secs(), 1e-9)
                steps = max(1, int(rate * o["trial_time"]))
            out.write(f"  Done in {num_str(time.perf_counter() - tw)} secs.\n")
        soln.clear_stats()
        first_t, last_t = t0, t0 + steps - 1
        out.write(DIV + f"Running {o['num_trials']} performance trial(s) of {steps} step(s) each...\n")
        trials = []
        for tr in range(o["num_trials"]):
            out.write(DIV + f"Trial number:  {tr + 1}\n")
            if o["validate"]:
                init_vars(soln)
            if o["sleep"] > 0:
                time.sleep(o["sleep"])
            env.global_barrier()
            soln.clear_stats()
            soln.run_solution(first_t, last_t)
            env.global_barrier()
            st = soln.get_stats()
            secs = st.get_elapsed_secs()
            n = st.get_num_steps_done()
            rec = {"nsteps": n, "run_time": secs, "pts_ps": st.get_num_elements() * n / secs,
                   "reads_ps": st.get_num_reads_done() / secs, "writes_ps": st.get_num_writes_done() / secs,
                   "flops": st.get_est_fp_ops_done() / secs}
            trials.append(rec)
            out.write(f" num-steps-done:           {n}\n elapsed-time (sec):       {num_str(secs)}\n"
                      f" throughput (num-points/sec): {num_str(rec['pts_ps'])}\n")
        trials.sort(key=lambda r: r["run_time"])
        best, mid = trials[0], trials[len(trials) // 2]
        pps = [t["pts_ps"] for t in trials]
        n = len(pps)
        ave = sum(pps) / n
        sd = math.sqrt(max(0.0, (sum(p * p for p in pps) - sum(pps) ** 2 / n) / (n - 1))) if n > 2 else 0.0
        out.write(DIV + "Throughput stats across trials:\n"
                  f" num-trials:                          {n}\n"
                  f" min-throughput (num-points/sec):     {num_str(min(pps))}\n"
                  f" max-throughput (num-points/sec):     {num_str(max(pps))}\n"
                  f" ave-throughput (num-points/sec):     {num_str(ave)}\n"
                  f" std-dev-throughput (num-points/sec): {num_str(sd)}\n")
        for tag, r, title in (("best", best, "best trial"), ("mid", mid, "50th-percentile trial")):
            pad = " " if tag == "mid" else ""
            out.write(DIV + f"Performance stats of {title}:\n"
                      f" {tag}-num-steps-done:              {pad}{r['nsteps']}\n"
                      f" {tag}-elapsed-time (sec):          {pad}{num_str(r['run_time'])}\n"
                      f" {tag}-throughput (num-reads/sec):  {pad}{num_str(r['reads_ps'])}\n"
                      f" {tag}-throughput (num-writes/sec): {pad}{num_str(r['writes_ps'])}\n"
                      f" {tag}-throughput (est-FLOPS):      {pad}{num_str(r['flops'])}\n"
                      f" {tag}-throughput (num-points/sec): {pad}{num_str(r['pts_ps'])}\n")
        ok = True
        if o["validate"]:
            out.write("\n" + DIV + "Setup for validation...\n")
            ref = fac.new_solution(env, soln)
            ref.apply_command_line_options(" ".join(rest))
            ref.apply_command_line_options("-force_scalar" + (" -no-overlap_comms -exchange_halos" if world > 1 else ""))
            ref.prepare_solution()
            init_vars(ref)
            out.write("\n" + DIV + f"Running {steps} step(s) for validation...\n")
            out.flush()
            ref.run_solution(first_t, last_t)      # same steps as the last (re-initialised) perf trial
            out.write(f"  Done in {num_str(ref.get_stats().get_elapsed_secs())} secs.\n\nChecking results...\n")
            errs = soln.compare_data(ref, 1e-3)
            for r in range(world):
                env.global_barrier()
                if r == rank:
                    if errs == 0:
                        sys.stderr.write(f"TEST PASSED on rank {rank}.\n")
                    else:
                        sys.stderr.write(f"TEST FAILED on rank {rank}: {errs} mismatch(es).\n")
                        ok = False
                    sys.stderr.flush()
            env.global_barrier()
            ref.end_solution()
        else:
            out.write("\nResults NOT VERIFIED.\n")
        soln.end_solution()
        out.write(f"Stencil '{soln.get_description()}'.\n")
        if not ok:
            return 1
        out.write("YASK DONE.\n" + DIV)
        out.flush()
        return 0
    except RuntimeError as e:
        sys.stderr.write(f"YASK Kernel: {e}.\n")
        return 1


if __name__ == "__main__":
    sys.exit(main())
