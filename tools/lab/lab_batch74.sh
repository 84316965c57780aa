#!/bin/bash
# round 3, batch 74: weight-stationary kernel v2 (one barrier per timestep, bias as C operand, hi operand read once, early rows)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab74; mkdir -p $O
V=build/variants
timeout 1200 python tools/tp_lab.py --rounds=3 r0=$V/libhns_ws2_r0.so r2=$V/libhns_ws2_r2.so 2>&1 | cut -c1-150 > $O/tp_lab.txt
cat $O/tp_lab.txt
