#!/bin/bash
# round 4, batch 75: bench.py with the configuration legs over 2 000 steps (400 before): the driver's command, the contract tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b75; mkdir -p $O
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>&1 | grep real; python tools/bench_line.py < $O/bench_driver.json | head -12
timeout 900 python -m pytest tests/test_bench_contract.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
