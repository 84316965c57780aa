#!/bin/bash
# round 3, batch 78: asynchronous history trim — tests; bench (default command) with both generator modes and the beyond-L3 leg
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab78; mkdir -p $O
timeout 900 python -m pytest tests/test_envgen.py tests/test_hip_envgen.py tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -8
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python tools/bench_line.py < $O/bench.json
