#!/bin/bash
# GPU job r5r: the added candidates (half-height plane-ring tiles for tti / test_scratch_3d, iso3dfd's star shape for iso3dfd_sponge):
# parity against the reference's outputs, then the shapes timed at 512^3.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5r; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 600 python3 -m pytest tests/test_reference_stencils_gpu.py tests/test_box_kernel_gpu.py -m gpu -q --timeout 500 -k "tti or test_scratch_3d or awp_abc or awp_elastic_abc or iso3dfd_sponge or picks_fast or ragged" 2>&1 | tail -6 > $O/parity.txt; tail -3 $O/parity.txt
timeout 300 python3 tools/sweep_variants.py --stencil tti --size 512 --reps 5 --chunks 0 --check --steps 10 --out $O/sweep_tti_p0.json > $O/sweep_tti.log 2>&1; grep -E "WHOLE|FAILED|mismatches vs naive: [1-9]" $O/sweep_tti.log | cut -c1-300; tail -1 $O/sweep_tti.log | cut -c1-900
timeout 300 python3 tools/sweep_variants.py --stencil iso3dfd_sponge --size 512 --reps 10 --chunks 0 --check --steps 20 --out $O/sweep_iso3dfd_sponge_p0.json > $O/sweep_sponge.log 2>&1; grep -E "WHOLE|FAILED|mismatches vs naive: [1-9]" $O/sweep_sponge.log | cut -c1-300; tail -1 $O/sweep_sponge.log | cut -c1-500
for p in 0 1 2; do timeout 200 python3 tools/sweep_variants.py --stencil test_scratch_3d --part $p --size 512 --reps 5 --chunks 0 $([ $p = 2 ] && echo "--steps 10") --out $O/sweep_test_scratch_3d_p$p.json > $O/sweep_ts3d_p$p.log 2>&1; grep WHOLE $O/sweep_ts3d_p$p.log | cut -c1-300; tail -1 $O/sweep_ts3d_p$p.log | cut -c1-400; done
