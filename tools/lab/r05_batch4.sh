#!/bin/bash
# round 5, batch 4: phase stamps of the farthest-point kernel's exchange
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_b4; mkdir -p $O
HNS_LIBRARY=build/variants/libhns_fpsph.so timeout 120 python tools/fps_phases.py 2>&1 | grep -v amdgpu.ids | tee $O/fps_phases.txt
