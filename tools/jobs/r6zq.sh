#!/bin/bash
# GPU job r6zq: awp_abc's one-plane parts on the vector point kernel with the lanes along y (tile 4 z x 256 y: z neighbours of a
# column from ONE 16-byte load instead of one 4-byte load each).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zq; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R YASK_HIP_LIB_DIR=$R/yask_amd/lib_x
for p in 1 4 5 6; do
  timeout 300 python3 tools/sweep_variants.py --stencil awp_abc --size 512 --part $p --chunks 0 --reps 5 --check --out $O/sweep_awp_abc_p$p.json > $O/sweep_awp_abc_p$p.log 2>&1
  echo "== awp_abc part $p"; grep "^{'variant'" $O/sweep_awp_abc_p$p.log | sed "s/'xchunk': 0, //; s/, 'gpoints.*//" | sort -t: -k3 -n | head -n 6; grep mismatches $O/sweep_awp_abc_p$p.log | sort | uniq -c | sort -rn | head -4
done
python3 tools/generic_table.py --out $O --only awp_abc awp_elastic_abc --size3 512 --tag yl 2>&1
