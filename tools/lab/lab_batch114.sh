#!/bin/bash
# round 3, batch 114: bench lines of the round's last build: default run, the driver's command, --traffic-live
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab114; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python tools/bench_line.py < $O/bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/bench_line.py < $O/bench_driver.json | head -1
timeout 600 python bench.py --traffic-live --no-cpu-baseline --config-steps 0 --tp-steps 0 --abi-steps 0 > $O/bench_live.json 2> $O/bench_live.err; python tools/bench_line.py < $O/bench_live.json | head -1
