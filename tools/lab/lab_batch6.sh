#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab6; mkdir -p $O
B=build/lab
L=$B/libhns_v2b_all.so
timeout 1200 python tools/step_lab.py --rounds=3 v2b=$L v2b_s1=$L:0:HNS_LAB_STAGGER=1 v2b_s2=$L:0:HNS_LAB_STAGGER=2 v2b_s3=$L:0:HNS_LAB_STAGGER=3 v2b_s4=$L:0:HNS_LAB_STAGGER=4 v2b_s6=$L:0:HNS_LAB_STAGGER=6 v2b_s8=$L:0:HNS_LAB_STAGGER=8 v2b_s12=$L:0:HNS_LAB_STAGGER=12 aux16_s4=$B/libhns_v2a_aux16.so:0:HNS_LAB_STAGGER=4 v2b_b=$L > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
