#!/bin/bash
# round 3, batch 69: weight-stationary kernel after the split fix: deviation per hidden unit, parameter ablations, A/B time, the TP tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab69; mkdir -p $O
V=build/variants
HNS_LIBRARY=$V/libhns_ws.so timeout 300 python tools/tp_debug2.py > $O/dbg2.txt 2>&1; grep -v amdgpu $O/dbg2.txt | cut -c1-200
HNS_LIBRARY=$V/libhns_ws.so timeout 300 python tools/tp_debug.py > $O/dbg.txt 2>&1; grep -v amdgpu $O/dbg.txt | cut -c1-250
timeout 900 python tools/tp_lab.py --rounds=3 ws=$V/libhns_ws.so > $O/tp_lab.txt 2>&1
HNS_TP_KERNEL=tile timeout 900 python tools/tp_lab.py --rounds=3 tile=$V/libhns_ws.so >> $O/tp_lab.txt 2>&1
cat $O/tp_lab.txt
HNS_LIBRARY=$V/libhns_ws.so timeout 600 python -m pytest tests/test_hip_tp.py -x -q -m gpu 2>&1 | tail -5
