#!/bin/bash
# GPU job r3c: one-step diagnostic of round 2's slab schedule at 1024^3 (1-ulp differences from the one-rank run); the decomposed-block tests.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== bit-exactness diagnostic, one step"; ( time timeout 600 python tools/diag_bitexact.py 1 legacy; python tools/diag_bitexact.py 2 legacy ) > $O/diag1.log 2>&1; echo "rc=$?"; grep -v "^Solution" $O/diag1.log | tail -40
