#!/bin/bash
# round 4, batch 74: no read-back at a masked reset while nothing consults the mirror: the boundary cost again, the GPU tests that cross episode boundaries
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b74; mkdir -p $O
B="python bench.py --steps 4000 --warmup 50 --no-traffic-live --config-steps 0 --tp-steps 0 --abi-steps 0 --no-cpu-baseline"
for rep in 1 2; do for ep in 800 1000000 200; do echo -n "episode $ep: "; timeout 300 $B --episode $ep 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'])"; done; done | tee $O/ep.txt
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_envgen.py tests/test_envgen.py tests/test_torchrl_branch.py tests/test_hip_fuzz.py tests/test_bench_contract.py tests/test_sharding_gloo.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
