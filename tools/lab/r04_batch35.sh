#!/bin/bash
# round 4, batch 35: role timeline of the tile mapping at the headline shape and the 6v2 shard
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b35; mkdir -p $O
timeout 200 python tools/phase_profile.py --envs=65536 --cylinders=8 --waves 2>&1 | grep -v amdgpu | tail -34 | tee $O/phase65536.txt
timeout 200 python tools/phase_profile.py --envs=65536 --cylinders=16 --agents=6 --targets=2 2>&1 | grep -v amdgpu | tail -18 | tee $O/phase65536_6v2.txt
