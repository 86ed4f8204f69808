#!/bin/bash
# GPU job r5x: iso3dfd_sponge with the one-row trip shape as a candidate: parity + shapes at 512^3.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5x; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 300 python3 -m pytest tests/test_reference_stencils_gpu.py -m gpu -q --timeout 300 -k "iso3dfd_sponge" 2>&1 | tail -n 2
timeout 300 python3 tools/sweep_variants.py --stencil iso3dfd_sponge --size 512 --reps 10 --chunks 0 --check --steps 20 --out $O/sweep_iso3dfd_sponge_p0.json > $O/sweep_sponge.log 2>&1; grep -E "WHOLE|FAILED|mismatches vs naive: [1-9]" $O/sweep_sponge.log | cut -c1-300; tail -n 1 $O/sweep_sponge.log | cut -c1-600
