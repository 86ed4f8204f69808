#!/bin/bash
# round 3, job j: with explicit FMAs -- timing of the _tl shapes against the plain ones (new code for both), the new bit-for-bit tests
cd /root/repo; mkdir -p gpurun_out/r3j
timeout 400 python tools/tail_probe.py > gpurun_out/r3j/tail_probe.txt 2>&1
timeout 600 python -m pytest tests/test_iso3dfd_gpu.py tests/test_stencils_gpu.py tests/test_decomposed_blocks_gpu.py -m gpu -x -q > gpurun_out/r3j/pytest.log 2>&1
grep -n "passed\|failed\|Error\|assert" gpurun_out/r3j/pytest.log | tail -8
grep -v "^Solution" gpurun_out/r3j/tail_probe.txt | cut -c1-260
