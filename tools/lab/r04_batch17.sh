#!/bin/bash
# round 4, GPU batch 17: final — the whole GPU suite, smoke, the driver's command, the default bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b17
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/r04b17/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/r04b17/pytest.log | cut -c1-260 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04b17/bench_driver.json 2> gpurun_out/r04b17/bench_driver.err ) 2>&1 | grep real
python tools/bench_line.py < gpurun_out/r04b17/bench_driver.json 2>&1 | head -3
( time timeout 400 python bench.py > gpurun_out/r04b17/bench_default.json 2> gpurun_out/r04b17/bench_default.err ) 2>&1 | grep real
python tools/bench_line.py < gpurun_out/r04b17/bench_default.json 2>&1 | head -20
