#!/bin/bash
# final state: full GPU suite, smoke, default bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab54; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; timeout 20 python tools/bench_line.py < $O/bench.json
