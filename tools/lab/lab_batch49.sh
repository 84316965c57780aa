#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab49; mkdir -p $O
B=build/lab
timeout 600 python tools/step_lab.py --rounds=3 final=$B/libhns_final_lab.so u8=$B/libhns_u8_lab.so u2=$B/libhns_u2_lab.so final_b=$B/libhns_final_lab.so u8_b=$B/libhns_u8_lab.so u2_b=$B/libhns_u2_lab.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
