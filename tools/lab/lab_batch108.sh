#!/bin/bash
# round 3, batch 108: phase stamps of the two-evader step kernel (config 5's shard) once more
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/phase_profile.py --agents=6 --cylinders=16 --targets=2 2>&1 | grep -v amdgpu | tail -22
timeout 300 python tools/phase_profile.py --agents=6 --cylinders=16 --targets=1 2>&1 | grep -v amdgpu | tail -22
