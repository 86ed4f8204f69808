#!/bin/bash
# GPU job r4b: the IPC transport with registrations exchanged once (host out of the exchange loop) -- transport / multirank /
# decomposed tests first, then the whole GPU suite, then the headline bench line.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_transport_gpu.py -m gpu -x -q 2>&1 | tail -40 > $O/transport.log; tail -5 $O/transport.log
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_transport_gpu.py --durations=15 2>&1 | tail -60 > $O/suite.log; tail -25 $O/suite.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-600
