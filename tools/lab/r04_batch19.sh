#!/bin/bash
# round 4, GPU batch 19: step kernel with the cylinder count as a compile-time constant for the reference's / BASELINE's shapes — whole suite + bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b19
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/r04b19/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/r04b19/pytest.log | cut -c1-260 | head -20
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04b19/bench_driver.json 2> gpurun_out/r04b19/bench_driver.err
python tools/bench_line.py < gpurun_out/r04b19/bench_driver.json 2>&1 | head -4
timeout 400 python bench.py > gpurun_out/r04b19/bench_default.json 2> gpurun_out/r04b19/bench_default.err
python tools/bench_line.py < gpurun_out/r04b19/bench_default.json 2>&1 | head -20
