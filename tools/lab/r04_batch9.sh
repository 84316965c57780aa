#!/bin/bash
# round 4, GPU batch 9: the whole GPU suite (step + predictor as half batches, final farthest-point policy, stats stride), then the bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b9
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r04b9/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/r04b9/pytest.log | cut -c1-260 | head -40
timeout 400 python bench.py > gpurun_out/r04b9/bench_default.json 2> gpurun_out/r04b9/bench_default.err
python tools/bench_line.py < gpurun_out/r04b9/bench_default.json 2>&1 | head -20; tail -3 gpurun_out/r04b9/bench_default.err
