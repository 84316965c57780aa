#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b21
timeout 900 python -m pytest tests/test_sharding_gloo.py tests/test_bench_contract.py tests/test_abi.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5
HNS_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 640 --warmup 64 --envs 32768 2>/dev/null | python -c "
import sys,json; d=json.load(sys.stdin); print('2 ranks x 32768 envs (gloo, one GPU):', d['ms_per_step'], d['value'], d['collective_us'])"
