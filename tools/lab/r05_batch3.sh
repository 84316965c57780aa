#!/bin/bash
# round 5, batch 3: the farthest-point trim with the points dealt round-robin over the workgroups and idle waves skipped: parity, time by XCD count
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_b3; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_envgen.py tests/test_envgen.py tests/test_bench_contract.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for x in "" 2 4; do
  echo "HNS_FPS_XCDS=$x" >> $O/fps_time.txt
  HNS_FPS_XCDS=$x timeout 120 python tools/fps_time.py 2>&1 | grep hns_fps >> $O/fps_time.txt
done
cat $O/fps_time.txt
