#!/bin/bash
# round 3, batch 113: the driver's command five times in fresh processes after moving the first event-bracketed launch and the event creation out of the timed region
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_bench_contract.py -q -x 2>&1 | tail -2
for i in 1 2 3 4 5; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python tools/bench_line.py | head -1
done
timeout 600 python bench.py --no-cpu-baseline --config-steps 0 --tp-steps 0 --abi-steps 0 2>/dev/null | python tools/bench_line.py | head -1
