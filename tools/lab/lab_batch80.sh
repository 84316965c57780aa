#!/bin/bash
# round 3, batch 80: two-evader step kernel — the pursuer lanes take over the order-free part of the evader policies: parity, phases, time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab80; mkdir -p $O
timeout 900 python -m pytest tests/test_two_evaders.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/phase_profile.py --agents=6 --cylinders=16 --targets=2 2>&1 | grep -v amdgpu | head -12
timeout 600 python bench.py --no-cpu-baseline --tp-steps 0 --abi-steps 0 --steps 200 --warmup 20 > $O/bench.json 2> $O/err; python tools/bench_line.py < $O/bench.json | grep "cfg5\|headline"
