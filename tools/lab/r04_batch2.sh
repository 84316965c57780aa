#!/bin/bash
# round 4, GPU batch 2: the GPU suite on the restructured step kernel (one design: tuned + generic instantiations), reference-mode
# defaults (pid_reset, reset_extra_step); the copy yardstick by kernel shape; the driver's bench command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b2
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r04b2/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r04b2/pytest.log | cut -c1-220 | head -60
timeout 120 ./build/microbench/copy_rate > gpurun_out/r04b2/copy_rate.txt 2>&1; cat gpurun_out/r04b2/copy_rate.txt
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04b2/bench_driver.json 2> gpurun_out/r04b2/bench_driver.err
python tools/bench_line.py < gpurun_out/r04b2/bench_driver.json 2>&1 | head -40
