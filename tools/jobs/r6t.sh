#!/bin/bash
# GPU job r6t: permuted (compile-time variant) solutions cut over ranks against the reference built with the same flags.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6t; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 1200 python3 -m pytest tests/test_compile_time_variants_gpu.py -m gpu -q --timeout 300 -k "ranks" 2>&1 | grep -v "^Solution '" ) > $O/tests.txt 2>&1
tail -n 40 $O/tests.txt
