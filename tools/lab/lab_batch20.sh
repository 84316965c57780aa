#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab20; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python tools/bench_line.py < $O/bench.json; python -c "
import json;d=json.load(open('$O/bench.json'));print(json.dumps({k:d[k] for k in ('value','ms_per_step','n_gpus','roofline','abi_rate','configs','tp_mode')},indent=1))"
