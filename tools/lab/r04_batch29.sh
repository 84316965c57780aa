#!/bin/bash
# round 4, batch 30: small mapping v4 (rows straight from registers, own LDS layout, env wave reads its cylinders from memory): parity, times, stamps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b30; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_reset_pid.py tests/test_hip_fuzz.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -6 ) 2>&1 | tail -10
for m in tile small; do echo "== mapping=$m"; HNS_STEP_MAPPING=$m timeout 200 python tools/small_batch.py 2048 4096 16384 32768 40960 49152 2>&1 | grep "E="; done 2>&1 | tee $O/ab.txt
timeout 200 python tools/phase_profile.py --envs=4096 --cylinders=5 --mapping=small --waves 2>&1 | grep -v amdgpu | tail -32 | tee $O/phase4096_small.txt
