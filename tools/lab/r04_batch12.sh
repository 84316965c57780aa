#!/bin/bash
# round 4, GPU batch 12: the whole GPU suite on the final sources, the driver's bench command and the default bench, the generator profile
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b12
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r04b12/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/r04b12/pytest.log | cut -c1-260 | head -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04b12/bench_driver.json 2> gpurun_out/r04b12/bench_driver.err
python tools/bench_line.py < gpurun_out/r04b12/bench_driver.json 2>&1 | head -4
timeout 400 python bench.py > gpurun_out/r04b12/bench_default.json 2> gpurun_out/r04b12/bench_default.err
python tools/bench_line.py < gpurun_out/r04b12/bench_default.json 2>&1 | head -20
timeout 500 bash tools/profile_envgen.sh r04_envgen
