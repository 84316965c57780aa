#!/bin/bash
# round 3, batch 79: final numbers — GPU suite, default bench, the driver's command, rocprofv3 profile of the headline step kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab79; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python tools/bench_line.py < $O/bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/bench_line.py < $O/bench_driver.json | head -3
timeout 600 bash tools/profile_step.sh r03 > $O/prof_step.txt 2>&1; tail -25 $O/prof_step.txt
