#!/bin/bash
# round 3, batch 105: final numbers of the round's last build — default bench, the driver's command, rocprofv3 stats of the default run's kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab105; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python tools/bench_line.py < $O/bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/bench_line.py < $O/bench_driver.json | head -1
timeout 600 bash tools/profile_step.sh r03_final > $O/prof.txt 2>&1; grep "step_v4" gpurun_out/prof_r03_final/stats.csv
