#!/bin/bash
# round 5, batch 6: generator with the host-side statistics fused; envgen / manifest / torchrl tests; bench legs
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_b6; mkdir -p $O
timeout 900 python -m pytest tests/test_envgen.py tests/test_hip_envgen.py tests/test_manifest.py tests/test_torchrl_branch.py tests/test_two_evaders.py tests/test_sharding_gloo.py tests/test_hip_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
EP_LEN=800 timeout 300 python tools/envgen_cost.py 2>&1 | grep -v amdgpu.ids | tee $O/envgen_cost.txt
timeout 600 python bench.py --steps 400 --warmup 100 --tp-steps 0 --abi-steps 0 --no-cpu-baseline --no-traffic-live > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python tools/bench_line.py < $O/bench.json
