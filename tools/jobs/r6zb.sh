#!/bin/bash
# GPU job r6zb: where does awp's marching kernel lose against ssg2's (same array count, 0.55 vs 0.70)?  Shape sweeps + SQ counters.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zb; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
for p in 0 1; do
  timeout 300 python3 tools/sweep_variants.py --stencil awp --size 512 --part $p --chunks 0 --reps 5 --out $O/sweep_awp_p$p.json > $O/sweep_awp_p$p.log 2>&1; tail -n 25 $O/sweep_awp_p$p.log
  timeout 300 python3 tools/sweep_variants.py --stencil awp_elastic --size 512 --part $p --chunks 0 --reps 5 --out $O/sweep_awpe_p$p.json > $O/sweep_awpe_p$p.log 2>&1; tail -n 25 $O/sweep_awpe_p$p.log
done
timeout 600 python3 tools/variant_pmc.py --stencil awp --variant march_v4_z128_y16_nt_w2 --part 0 --out $O/pmc_awp_p0 > $O/pmc_awp_p0.log 2>&1; tail -n 30 $O/pmc_awp_p0.log
timeout 600 python3 tools/variant_pmc.py --stencil awp --variant march_v2_z128_y8_w2 --part 1 --out $O/pmc_awp_p1 > $O/pmc_awp_p1.log 2>&1; tail -n 30 $O/pmc_awp_p1.log
timeout 600 python3 tools/variant_pmc.py --stencil ssg2 --variant march_v4_z128_y16_nt_hr_ps_w2 --part 0 --out $O/pmc_ssg2_p0 > $O/pmc_ssg2_p0.log 2>&1; tail -n 30 $O/pmc_ssg2_p0.log
