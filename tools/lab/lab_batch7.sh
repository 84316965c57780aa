#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab7; mkdir -p $O
B=build/lab
timeout 1200 python tools/step_lab.py --rounds=3 v2b=$B/libhns_v2b_all.so noslp=$B/libhns_v2b_noslp.so maxilp=$B/libhns_v2b_maxilp.so minreg=$B/libhns_v2b_minreg.so v2b_b=$B/libhns_v2b_all.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
