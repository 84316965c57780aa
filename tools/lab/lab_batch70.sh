#!/bin/bash
# round 3, batch 70: phase profile of the weight-stationary kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab70; mkdir -p $O
HNS_LIBRARY=build/variants/libhns_ws_ph.so timeout 300 python tools/tp_phase_profile.py --ws > $O/phases.txt 2>&1
grep -v amdgpu $O/phases.txt
