#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab36; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python bench.py --no-cpu-baseline --tp-steps 0 --stream-groups 0 --config-steps 100 --abi-steps 200 > $O/bench.json 2> $O/bench.err; python tools/bench_line.py $O/bench.json 2>/dev/null || cut -c1-600 $O/bench.json
