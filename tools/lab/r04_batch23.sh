#!/bin/bash
# round 4, batch 23: the small-batch mapping (hns_step_small_kernel): parity suite with the automatic choice, step times tile vs small, phase stamps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b23; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_reset_pid.py tests/test_hip_fuzz.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -6 ) 2>&1 | tail -10
for m in tile small; do echo "== $m"; HNS_STEP_MAPPING=$m timeout 200 python tools/small_batch.py 1024 2048 4096 8192 16384 32768 65536 2>&1 | grep "E=" | tee $O/small_$m.txt; done
timeout 200 python tools/phase_profile.py --envs=4096 --cylinders=5 --mapping=small 2>&1 | grep -v amdgpu | tail -16 | tee $O/phase4096_small.txt
timeout 200 python tools/phase_profile.py --envs=4096 --cylinders=5 --mapping=tile 2>&1 | grep -v amdgpu | tail -16 | tee $O/phase4096_tile.txt
