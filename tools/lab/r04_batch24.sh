#!/bin/bash
# round 4, batch 24: small mapping, per-wave stamps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b24; mkdir -p $O
timeout 200 python tools/phase_profile.py --envs=4096 --cylinders=5 --mapping=small --waves 2>&1 | grep -v amdgpu | tail -32 | tee $O/phase4096_small.txt
