#!/bin/bash
# round 5, batch 10: the 16-candidate trim in the env: parity (generator + whole suite parts), generator cost, fuzz of the generator shapes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_b10; mkdir -p $O
HNS_FUZZ_GEN_SEEDS=300 timeout 900 python -m pytest tests/test_hip_envgen.py tests/test_envgen.py tests/test_hip_fuzz.py tests/test_two_evaders.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for x in 1 2 4; do HNS_FPS_XCDS=$x timeout 300 python -m pytest tests/test_hip_envgen.py -m gpu -x -q 2>&1 | tail -1; done
for b in 1 2 3 5 8 12; do HNS_FPS_BATCH=$b timeout 300 python -m pytest tests/test_hip_envgen.py -m gpu -x -q 2>&1 | tail -1; done
EP_LEN=800 timeout 300 python tools/envgen_cost.py 2>&1 | grep -v amdgpu.ids | tee $O/envgen_cost.txt
timeout 120 python tools/fps_time.py 2>&1 | grep hns_fps | tee $O/fps_time.txt
