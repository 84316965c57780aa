#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b16
timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6
( time timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04b16/bench_driver.json 2> gpurun_out/r04b16/bench_driver.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('gpurun_out/r04b16/bench_driver.json')); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['traffic'], r['traffic_source'][:60])"
timeout 200 python examples/rollout.py 2>&1 | tail -5
