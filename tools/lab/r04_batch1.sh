#!/bin/bash
# round 4, GPU batch 1: the whole GPU suite, the driver's bench command, the default bench, the step-kernel profile
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b1
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r04b1/pytest.log 2>&1
tail -3 gpurun_out/r04b1/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04b1/bench_driver.json 2> gpurun_out/r04b1/bench_driver.err
cut -c1-900 gpurun_out/r04b1/bench_driver.json
timeout 400 python bench.py > gpurun_out/r04b1/bench_default.json 2> gpurun_out/r04b1/bench_default.err
cut -c1-900 gpurun_out/r04b1/bench_default.json
timeout 400 bash tools/profile_step.sh r04
