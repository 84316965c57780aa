#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab4; mkdir -p $O
B=build/lab
timeout 1200 python tools/step_lab.py --rounds=3 r1=$B/libhns_r1.so aux0=$B/libhns_v2a_aux0.so aux2_nt=$B/libhns_v2a_aux2.so aux16_sc1=$B/libhns_v2a_aux16.so aux17=$B/libhns_v2a_aux17.so aux18=$B/libhns_v2a_aux18.so v2a=$B/libhns_v2a.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
