#!/bin/bash
# full GPU suite + default bench + rocprofv3 profile of the step kernel (r02 final)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab40; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; timeout 20 python tools/bench_line.py < $O/bench.json
timeout 900 bash tools/profile_step.sh r02_v4 > $O/profile.log 2>&1; tail -30 $O/profile.log
