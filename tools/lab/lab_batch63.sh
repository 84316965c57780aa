#!/bin/bash
# round 3, batch 63: weight-stationary kernel, cross terms of every chunk before the leading terms
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab63; mkdir -p $O
V=build/variants
timeout 900 python tools/tp_lab.py --rounds=2 ws=$V/libhns_ws.so > $O/tp_lab.txt 2>&1
cat $O/tp_lab.txt
