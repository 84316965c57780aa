#!/bin/bash
# round 4, batch 48: bench.py with the copy yardstick ahead of the warm-up and the priority boost: the driver's command three times, the default once, the contract tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b48; mkdir -p $O
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 --no-traffic-live --config-steps 0 --tp-steps 0 --abi-steps 0 --no-cpu-baseline > $O/drv$i.json 2>$O/drv$i.err; python tools/bench_line.py < $O/drv$i.json | head -1; done
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/bench_line.py < $O/bench_driver.json | head -12
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_line.py < $O/bench_default.json | head -12
timeout 900 python -m pytest tests/test_bench_contract.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
