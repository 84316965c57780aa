#!/bin/bash
# round 3, batch 92: verification of the product build — whole GPU suite, smoke(), default bench, the driver's command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab92; mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python tools/bench_line.py < $O/bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/bench_line.py < $O/bench_driver.json | head -1
