#!/bin/bash
# round 3, batch 111: headline leg three times in fresh processes (the first masked reset is primed before the warm-up now)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline --config-steps 0 --tp-steps 0 --abi-steps 0 2>/dev/null | python tools/bench_line.py | head -1
done
