#!/bin/bash
# GPU job r3d: multi-rank wave-front tiling tests; overlap probe (mirror transport); decomposed-block tests again; headline bench.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== wavefront multi-rank tests"; ( time timeout 300 python -m pytest tests/test_transport_gpu.py -q --timeout 120 ) > $O/pytest_wf.log 2>&1; echo "rc=$?"; tail -25 $O/pytest_wf.log
echo "== overlap probe iso3dfd"; ( time timeout 300 python tools/overlap_probe.py --stencil iso3dfd ) > $O/overlap_iso3dfd.log 2>&1; echo "rc=$?"; python - <<'PY'
import json
try:
    for r in json.load(open("gpurun_out/overlap_probe_iso3dfd.json")):
        print(r["case"][:30].ljust(30), r["schedule"][:30].ljust(30), "ms", r["ms_per_step"], "vs1", r["vs_one_rank_block"], "ext", r["exterior_ms"], "int", r["interior_ms"], "pack", r["pack_ms"], "copy", r["copy_ms"], "unpack", r["unpack_ms"], "wait", r["exposed_wait_ms"], "hidden", r["comm_hidden_fraction"], "MB", r["halo_MB_per_step"])
except Exception as e:
    print("no result", e)
PY
tail -5 $O/overlap_iso3dfd.log
echo "== overlap probe ssg"; ( time timeout 300 python tools/overlap_probe.py --stencil ssg ) > $O/overlap_ssg.log 2>&1; echo "rc=$?"; python - <<'PY'
import json
try:
    for r in json.load(open("gpurun_out/overlap_probe_ssg.json")):
        print(r["case"][:30].ljust(30), r["schedule"][:30].ljust(30), "ms", r["ms_per_step"], "vs1", r["vs_one_rank_block"], "ext", r["exterior_ms"], "int", r["interior_ms"], "pack", r["pack_ms"], "copy", r["copy_ms"], "unpack", r["unpack_ms"], "wait", r["exposed_wait_ms"], "hidden", r["comm_hidden_fraction"], "MB", r["halo_MB_per_step"])
except Exception as e:
    print("no result", e)
PY
cp gpurun_out/overlap_probe_*.json $O/ 2>/dev/null
echo "== decomposed blocks"; ( time timeout 600 python -m pytest tests/test_decomposed_blocks_gpu.py -q --timeout 300 ) > $O/pytest_blocks.log 2>&1; echo "rc=$?"; tail -6 $O/pytest_blocks.log

