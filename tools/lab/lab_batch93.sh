#!/bin/bash
# round 3, batch 93: phase stamps of the step kernel at 4 096 envs (BASELINE config 2: one workgroup per four CUs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/phase_profile.py --envs=4096 --cylinders=5 2>&1 | grep -v amdgpu | tail -30
timeout 300 python tools/phase_profile.py --envs=65536 --cylinders=5 2>&1 | grep -v amdgpu | tail -30
