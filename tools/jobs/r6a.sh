#!/bin/bash
# GPU job r6a: (1) the round's new parity tests -- multi-tile reference fixtures for every generic 3-D solution, the reference matrix's
# compile-time variants; (2) dry run of the bounded multi-device matrix with its wall time; (3) default bench; (4) iso3dfd over-fetch:
# XCD strips vs 4x2 / 2x4 XCD blocks (YKH_PROFILING library), FETCH_SIZE + kernel time per map.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6a; mkdir -p $O; cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$R
( time timeout 900 python3 -m pytest tests/test_multi_tile_fixtures_gpu.py tests/test_compile_time_variants_gpu.py -m gpu -q --timeout 300 -rs -s 2>&1 ) > $O/new_tests.txt 2>&1
tail -n 40 $O/new_tests.txt
( time YASK_TEST_MULTI_DEVICE_DRYRUN=1 timeout 600 python3 -m pytest tests/test_multi_device_gpu.py -m gpu -q --timeout 300 --durations=0 2>&1 ) > $O/first_contact_dryrun.txt 2>&1
tail -n 30 $O/first_contact_dryrun.txt
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
cd /tmp
V=starlin_v4_z128_y32_r2_t2_nt_pd2_tl_w2_c2
for m in 0 1 2; do
  for pass in FETCH_SIZE WRITE_SIZE; do
    YASK_HIP_LIB_DIR=$R/yask_amd/lib_prof YASK_HIP_XCD_MAP=$m timeout 300 rocprofv3 --kernel-trace --pmc $pass -f csv -d $O/xcd_map$m/$pass -- \
      python3 $R/bench.py --no-cpu-baseline --no-probe --traffic none --steps 6 --warmup 2 --ramp-secs 0.5 --opts "-hip_variant $V" > $O/xcd_map${m}_$pass.log 2>&1
  done
  YASK_HIP_LIB_DIR=$R/yask_amd/lib_prof YASK_HIP_XCD_MAP=$m timeout 300 python3 $R/bench.py --no-cpu-baseline --no-probe --traffic none --steps 40 --warmup 5 --opts "-hip_variant $V" > $O/xcd_map${m}_bench.json 2>/dev/null
done
python3 - <<PY
import csv, glob, collections, json
for m in (0, 1, 2):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/xcd_map%d/**/*counter_collection.csv" % m, recursive=True):
        for r in csv.DictReader(open(f)):
            if "starlin" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    avg = {k: sum(v) / len(v) for k, v in acc.items()}
    try:
        j = json.loads([l for l in open("$O/xcd_map%d_bench.json" % m) if l.startswith("{")][0])
        ms = j["roofline"]["kernel_ms"]
    except Exception as e:
        ms = repr(e)
    print("xcd_map", m, "fetch_GB(x2)=%.3f" % (avg.get("FETCH_SIZE", 0) * 1024 * 2 / 1e9), "write_GB=%.3f" % (avg.get("WRITE_SIZE", 0) * 1024 / 1e9), "n=%d" % len(acc.get("FETCH_SIZE", [])), "kernel_ms", ms)
PY
