#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab13; mkdir -p $O
B=build/lab
timeout 1200 python tools/step_lab.py --rounds=3 v3c=$B/libhns_v3c.so recsc1=$B/libhns_v3c_recsc1.so allplain=$B/libhns_v3c_allplain.so v3c_b=$B/libhns_v3c.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
