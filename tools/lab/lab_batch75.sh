#!/bin/bash
# round 3, batch 75: product build — TP tests, rocprofv3 profile of the predictor path (stats + counters), phase profile
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab75; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_tp.py tests/test_manifest.py -x -q -m gpu 2>&1 | tail -3
timeout 500 bash tools/profile_tp.sh r03 > $O/prof.txt 2>&1; tail -30 $O/prof.txt
HNS_LIBRARY=build/variants/libhns_ws_ph.so timeout 300 python tools/tp_phase_profile.py --ws 2>&1 | grep -v amdgpu | tail -14
