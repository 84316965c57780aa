#!/bin/bash
# round 4, batch 77: bench lines of the round's last build (configuration legs over 2 000 steps, no read-back at episode boundaries)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b77; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/bench_line.py < $O/bench_driver.json | head -12
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_line.py < $O/bench_default.json | head -12
