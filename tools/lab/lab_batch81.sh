#!/bin/bash
# round 3, batch 81: final verification of the product build — GPU suite, smoke(), bench, profile of the two-evader kernel after the rebalancing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab81; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python tools/bench_line.py < $O/bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/bench_line.py < $O/bench_driver.json | head -1
timeout 600 bash tools/profile_step.sh r03_a6t2 --agents 6 --targets 2 --cylinders 16 > $O/prof_a6t2.txt 2>&1; grep "step_v4" gpurun_out/prof_r03_a6t2/stats.csv
