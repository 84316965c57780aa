#!/bin/bash
# round 5, batch 8: phase stamps of the 6v2 step kernel and of the 3v1 kernel for comparison
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_b8; mkdir -p $O
timeout 120 python tools/phase_profile.py --agents=6 --cylinders=16 --targets=2 2>&1 | grep -v amdgpu.ids | tee $O/phase_a6t2.txt
timeout 120 python tools/phase_profile.py 2>&1 | grep -v amdgpu.ids | tee $O/phase_a3t1.txt
