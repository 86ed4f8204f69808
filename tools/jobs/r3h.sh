#!/bin/bash
# round 3, job h: where does the slab schedule with a split interior differ from the one-rank run? (per-plane histogram, pinned x-chunks)
cd /root/repo; mkdir -p gpurun_out/r3h
( time timeout 420 python tools/diag_bitexact2.py ) > gpurun_out/r3h/diag2.txt 2>&1
tail -30 gpurun_out/r3h/diag2.txt
