#!/bin/bash
# round 3, batch 71: weight-stationary kernel — window-row prefetch moved off the frame phase; issue-priority rotation variants; placement of co-resident workgroups
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab71; mkdir -p $O
V=build/variants
HNS_LIBRARY=$V/libhns_ws_ph.so timeout 300 python tools/tp_phase_profile.py --ws > $O/phases.txt 2>&1
grep -v amdgpu $O/phases.txt | tail -13
timeout 1200 python tools/tp_lab.py --rounds=3 p0=$V/libhns_ws_p0.so p1=$V/libhns_ws_p1.so p2=$V/libhns_ws_p2.so p3=$V/libhns_ws_p3.so p0_b=$V/libhns_ws_p0.so > $O/tp_lab.txt 2>&1
cat $O/tp_lab.txt
