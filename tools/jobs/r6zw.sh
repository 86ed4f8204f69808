#!/bin/bash
# GPU job r6zw: the generic marching solutions at 1024^3 (awp: 38 arrays x 4.3 GB = 163 GB resident of the 288 GB).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6zw; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
timeout 1200 python3 tools/generic_table.py --out $O --only awp awp_elastic ssg2 iso3dfd_sponge --size3 1024 --steps 20 --tag big 2>&1
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "total memory" | head -2
