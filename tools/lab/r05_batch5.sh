#!/bin/bash
# round 5, batch 5: farthest-point trim with two candidates + a bound per workgroup: parity, time, phase stamps
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_b9; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_envgen.py tests/test_envgen.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for b in 16 8 4; do
  echo "HNS_FPS_BATCH=$b" >> $O/fps_time.txt
  HNS_FPS_BATCH=$b timeout 120 python tools/fps_time.py 2>&1 | grep hns_fps >> $O/fps_time.txt
done
cat $O/fps_time.txt
HNS_LIBRARY=build/variants/libhns_fpsph.so timeout 120 python tools/fps_phases.py 2>&1 | grep -v amdgpu.ids | tee $O/fps_phases.txt
