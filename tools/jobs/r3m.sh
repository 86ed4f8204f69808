#!/bin/bash
# GPU job r3m: which compiled iso3dfd shape / x-chunk is fastest on the blocks a rank gets under strong scaling of the 1024^3 grid?
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r3m; mkdir -p $O; cd $R
timeout 300 python tools/sweep_variants.py --stencil iso3dfd --size 512 --chunks 0 256 128 64 --reps 20 --out $O/sweep_512.json > $O/sweep_512.log 2>&1
timeout 300 python tools/sweep_variants.py --stencil iso3dfd --size 512 512 1024 --chunks 0 256 128 --reps 12 --out $O/sweep_512x512x1024.json > $O/sweep_512x512x1024.log 2>&1
timeout 300 python tools/sweep_variants.py --stencil iso3dfd --size 512 1024 1024 --chunks 0 512 256 --reps 8 --out $O/sweep_512x1024x1024.json > $O/sweep_512x1024x1024.log 2>&1
for f in 512 512x512x1024 512x1024x1024; do echo "== $f"; grep -v "^Solution" $O/sweep_$f.log | sort -t, -k3 -n | head -0; python - "$O/sweep_$f.json" <<'PY'
import json,sys
try:
    rs=json.load(open(sys.argv[1]))
    rs=rs["results"] if isinstance(rs,dict) else rs
    rs=sorted(rs,key=lambda r:r["ms"])[:14]
    for r in rs: print(r.get("variant"), r.get("xchunk"), r["ms"], r.get("gpoints_per_s"))
except Exception as e:
    print("ERR",e); print(open(sys.argv[1].replace(".json",".log")).read()[-1500:])
PY
done
