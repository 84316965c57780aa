#!/bin/bash
# final state of round 2: full GPU suite, smoke, default bench, 2-rank bench, rocprofv3 profile of the step kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab47; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; timeout 20 python tools/bench_line.py < $O/bench.json
HNS_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 400 --warmup 50 > $O/bench_2ranks.json 2> $O/bench_2.err; timeout 20 python tools/bench_line.py < $O/bench_2ranks.json
timeout 900 bash tools/profile_step.sh r02_final > $O/profile.log 2>&1; tail -4 $O/profile.log
