#!/bin/bash
# round 4, batch 69: final verification of the round's last build: whole GPU suite, smoke(), soak (6 000 steps in four modes), the driver's command, the default bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b69; mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4 ) 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
timeout 600 python tools/soak.py 6000 2>&1 | grep -v amdgpu | tail -6
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/bench_line.py < $O/bench_driver.json | head -12
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_line.py < $O/bench_default.json | head -12
