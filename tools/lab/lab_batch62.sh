#!/bin/bash
# round 3, batch 62: weight-stationary predictor kernel (first version) against the tile kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab62; mkdir -p $O
V=build/variants
timeout 900 python tools/tp_lab.py --rounds=3 ws=$V/libhns_ws.so > $O/tp_lab.txt 2>&1
HNS_TP_KERNEL=tile timeout 900 python tools/tp_lab.py --rounds=3 tile=$V/libhns_ws.so >> $O/tp_lab.txt 2>&1
cat $O/tp_lab.txt
HNS_LIBRARY=$V/libhns_ws.so timeout 600 python -m pytest tests/test_hip_tp.py -x -q -m gpu 2>&1 | tail -15
