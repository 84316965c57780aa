#!/bin/bash
# round 4, batch 33: whole GPU suite + smoke + bench on the build with both step mappings
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b33; mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -6 ) 2>&1 | tail -10
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/bench_line.py < $O/bench_driver.json | head -30
