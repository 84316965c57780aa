#!/bin/bash
# round 3, batch 110: bench.py --traffic-live (roofline.traffic from two rocprofv3 PMC passes of the run itself)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab110; mkdir -p $O
( time timeout 900 python bench.py --traffic-live --no-cpu-baseline --config-steps 0 --tp-steps 0 --abi-steps 0 > $O/bench_live.json 2> $O/bench_live.err ) 2>&1 | tail -3
tail -3 $O/bench_live.err; python -c "
import json; r=json.load(open('$O/bench_live.json')); print(r['value'], r['roofline']['traffic'], r['roofline'].get('traffic_detail'), r['roofline']['traffic_source'][:60])"
