#!/bin/bash
# round 4, batch 40: which workgroups of a 65 536-env launch are slow (the drain of the single residency round)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b40; mkdir -p $O
for i in 1 2; do timeout 200 python tools/phase_profile.py --envs=65536 --cylinders=8 --spread 2>&1 | grep -v amdgpu | tail -9; done | tee $O/spread.txt
