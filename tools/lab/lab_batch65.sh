#!/bin/bash
# round 3, batch 65: weight-stationary kernel with the explicit B-operand ring: deviation from the oracle, time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab65; mkdir -p $O
V=build/variants
HNS_LIBRARY=$V/libhns_ws.so timeout 300 python tools/tp_debug.py > $O/dbg.txt 2>&1
cat $O/dbg.txt | cut -c1-250
timeout 900 python tools/tp_lab.py --rounds=3 ws=$V/libhns_ws.so > $O/tp_lab.txt 2>&1
HNS_TP_KERNEL=tile timeout 900 python tools/tp_lab.py --rounds=3 tile=$V/libhns_ws.so >> $O/tp_lab.txt 2>&1
cat $O/tp_lab.txt
