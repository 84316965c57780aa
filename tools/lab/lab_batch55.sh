#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab55; mkdir -p $O
B=build/lab
timeout 700 python tools/step_lab.py --rounds=3 final=$B/libhns_final_lab.so maxilp=$B/libhns_maxilp_lab.so itilp=$B/libhns_itilp_lab.so itocc=$B/libhns_itocc_lab.so final_b=$B/libhns_final_lab.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
