#!/usr/bin/env python
"""bench.py -- headline benchmark: iso3dfd 16th-order fp32, 1024^3 points per GPU (BASELINE.json configs[1]).

A "step" is one pass of the hot path (yk_solution::run_solution over one time step: one HIP stencil
launch per part, plus halo exchange when N>1) over the whole grid.  Metric: Gpoints/s =
overall_domain_points * steps / elapsed (the reference's "throughput (num-points/sec)",
src/kernel/lib/soln_apis.cpp:455-461), with all vars already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N>1 is launched by the driver as `python -m torch.distributed.run ... bench.py --gpus N`; the grid is
decomposed in x (one 1024^3 block per GPU, weak scaling), halos travel as RCCL send/recv on a side
stream overlapped with the interior kernel.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)

# workload -> (stencil library, description, default points per GPU per dim, dtype, algorithmic bytes per
# point-step, init: var -> (offset, scale, hash id))
WORKLOADS = {
    # BASELINE.json configs[1] -- the headline: read p(t) 4 + p(t-1) 4 + v 4, write p(t+1) 4
    "iso3dfd": ("iso3dfd", "iso3dfd r=8 fp32", 1024, "f32", 16.0, {"p": (0.0, 1.0, 0), "v": (150.0, 50.0, 1)}),
    # configs[2]: read 8 + write 8
    "3axis": ("3axis", "3axis r=4 fp64", 512, "f64", 16.0, {"A": (0.0, 1.0, 0)}),
    "heat3d": ("3axis_r1", "3axis r=1 (classic 7-point heat3d) fp64", 512, "f64", 16.0, {"A": (0.0, 1.0, 0)}),
    # configs[4]: stage 1 (10 reads + 3 writes) + stage 2 (12 reads + 6 writes), 4 B each
    "ssg": ("ssg", "ssg staggered-grid elastic fp32 (2 stages)", 512, "f32", 124.0,
            {**{f: (0.0, 1.0e-3, i + 1) for i, f in enumerate(["v_bl_w", "v_tl_v", "v_tr_u", "s_bl_yz", "s_br_xz", "s_tl_xx",
                                                              "s_tl_yy", "s_tl_zz", "s_tr_xy"])},
             "rho": (1.5, 0.5, 0), "mu": (1.5, 0.5, 10), "lambda": (1.5, 0.5, 11), "lambdamu2": (1.5, 0.5, 12)}),
}


def host_cores():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a container can see
    256 logical CPUs and be allowed 16 of them; 256 spinning OpenMP threads on 16 CPUs would time the scheduler)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = max(1, min(n, int(float(quota) / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(n=512, steps=10):
    """Reference CPU kernel (oracle/_ref, unmodified intel/yask) timed on this host's cores on a
    bounded sample of the same workload; falls back to the C restatement (kind 'port').  The thread count is
    the best of a short sweep (all usable CPUs, half of them = one per core with SMT, a quarter)."""
    cores = host_cores()
    flags = open("/proc/cpuinfo").read()
    arch = "avx512" if "avx512f" in flags else "avx2"
    exe = ROOT / "oracle" / "_ref" / "bin" / f"ref_driver.iso3dfd.{arch}.exe"
    sample = f"iso3dfd r=8 fp32 {n}^3 x {steps} steps (same stencil, 1/8 of the grid)"
    if exe.exists():
        best = None
        t_start = time.time()
        for thr in sorted({cores, max(1, cores // 2), max(1, cores // 4)}, reverse=True):
            if best is not None and time.time() - t_start > 60:
                break
            try:
                out = subprocess.run([str(exe), "-g", str(n), "-steps", str(steps), "-threads", str(thr), "-trials", "2",
                                      "-init", "v:150:50"], capture_output=True, text=True, timeout=300)
                for line in out.stdout.splitlines():
                    if line.startswith("{"):
                        j = json.loads(line)
                        if best is None or j["gpoints_per_s"] > best["gpoints_per_s"]:
                            best = j
            except Exception as e:  # noqa: BLE001
                print("cpu_baseline: reference run failed:", e, file=sys.stderr)
        if best is not None:
            return {"value": round(best["gpoints_per_s"], 4), "unit": "Gpoints/s", "cores": int(best["threads"]),
                    "kind": "reference", "sample": sample + f", yask target {best['target']}, best of 2 trials, best thread count of "
                                                             f"{cores}/{max(1, cores // 2)}/{max(1, cores // 4)} ({os.cpu_count()} logical CPUs visible)"}
    from oracle import oracle as O
    import numpy as np
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    n2, st2 = 256, 4
    t0 = time.time()
    O.run_iso3dfd((n2, n2, n2), st2, dtype=np.float32)
    dt = time.time() - t0
    return {"value": round(n2 ** 3 * st2 / dt * 1e-9, 4), "unit": "Gpoints/s", "cores": cores, "kind": "port",
            "sample": f"C restatement (oracle/stencil_oracle.c, OpenMP) {n2}^3 x {st2} steps incl. init"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="iso3dfd", choices=sorted(WORKLOADS), help="default: the headline (BASELINE.json configs[1])")
    ap.add_argument("--size", type=int, default=0, help="points per GPU in each dim (default: the workload's configured size)")
    ap.add_argument("--decomp", default="xslab", choices=["xslab", "compact"],
                    help="xslab (default): N x 1 x 1 ranks, contiguous whole-plane faces, 2 neighbours per GPU; compact: the "
                         "reference's default most-compact rank grid (8 -> 2x2x2), as in BASELINE.json configs[3]")
    ap.add_argument("--points-per-gpu", dest="local", type=int, nargs=3, default=None, metavar=("NX", "NY", "NZ"),
                    help="points per GPU per dim (overrides --size), e.g. --decomp compact --points-per-gpu 1024 1024 512")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "torch"])
    ap.add_argument("--opts", default="", help="extra yask options, e.g. '-hip_variant NAME'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # yask options start with '-': hand "--opts '-hip_variant X'" to argparse as "--opts=-hip_variant X"
    argv = sys.argv[1:]
    for i in range(len(argv) - 1):
        if argv[i] == "--opts":
            argv[i:i + 2] = ["--opts=" + argv[i + 1]]
            break
    args = ap.parse_args(argv)

    import torch
    from yask_amd import yk_factory, dist as ydist

    rank, local_rank, world = ydist.init_process_group()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    stencil, descr, dflt_n, dtype, BYTES_PER_POINT, init = WORKLOADS[args.workload]
    if local_rank == 0:
        from yask_amd import _capi
        _capi.ensure_built((stencil,))     # no-op when the kernel library is already built in-tree
    if world > 1:
        torch.distributed.barrier()
    fac = yk_factory(stencil)
    env, transport = ydist.new_env(fac, args.transport)
    soln = fac.new_solution(env)
    n = args.size or dflt_n
    local = list(args.local) if args.local else [n, n, n]
    if args.decomp == "xslab":
        # x-slab decomposition: x-faces are whole contiguous planes and each GPU has only 2 neighbours
        soln.set_num_ranks_vec([world, 1, 1])
    soln.set_rank_domain_size_vec(local)
    if args.opts:
        rem = soln.apply_command_line_options(args.opts)
        assert rem == "", rem
    soln.prepare_solution()
    for name, (off, sc, hid) in init.items():
        soln.get_var(name).set_elements_hash(off, sc, hash_id=hid)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    t = 0
    if args.warmup > 0:
        soln.run_solution(t, t + args.warmup - 1)
        t += args.warmup
    soln.get_stats()
    barrier()
    t0 = time.perf_counter()
    soln.run_solution(t, t + args.steps - 1)      # returns after the streams have drained
    barrier()
    elapsed = time.perf_counter() - t0
    t += args.steps
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    pts_per_gpu = float(local[0]) * local[1] * local[2]
    grid = soln.get_num_ranks_vec()
    total_pts = pts_per_gpu * world
    value = total_pts * args.steps / elapsed * 1e-9

    # dominant kernel(s): average launch duration by HIP events on the compute stream, same launches
    # (multi-stage solutions: the sum over their parts = one step's worth of launches)
    nparts = soln.get_num_parts()
    kern_ms = sum(soln.time_part(part=p, variant=-1, t=t, reps=max(10, min(args.steps, 50))) for p in range(nparts))
    achieved = BYTES_PER_POINT * pts_per_gpu / (kern_ms * 1e-3) * 1e-9
    traffic = None
    tf = ROOT / "profiles" / "hbm_traffic.json"
    if tf.exists() and args.workload == "iso3dfd" and local == [1024, 1024, 1024]:
        try:
            traffic = json.load(open(tf)).get("iso3dfd_1024_bytes_per_launch")
        except Exception:  # noqa: BLE001
            traffic = None

    if rank == 0:
        out = {
            "metric": "Gpoints/s (grid updates/s), " + ("iso3dfd 16th-order fp32" if args.workload == "iso3dfd" else descr),
            "value": round(value, 3), "unit": "Gpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic (logical-index hash init, random-like)",
            "config": {"workload": f"{descr}, {local[0]}x{local[1]}x{local[2]} points per GPU" if args.local else
                                   f"{descr}, {n}^3 points per GPU, global {n * world}x{n}x{n}",
                       "decomposition": (f"x-slabs {world}x1x1" if args.decomp == "xslab" else
                                         "rank grid " + "x".join(str(g) for g in grid) + ", global " +
                                         "x".join(str(g * l) for g, l in zip(grid, local))),
                       "halo_transport": transport,
                       "kernel": "+".join(soln.get_kernel_variant(p) for p in range(nparts)), "overlap_comms": True},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": BYTES_PER_POINT * pts_per_gpu},
            "gpoints_per_s_per_gpu": round(value / world, 3),
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "iso3dfd":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
