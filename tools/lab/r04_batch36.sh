#!/bin/bash
# round 4, batch 36: phase stamps of the predictor kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b36; mkdir -p $O
timeout 200 python tools/tp_phases.py 65536 2>&1 | grep -v amdgpu | tail -12 | tee $O/tp_phases.txt
