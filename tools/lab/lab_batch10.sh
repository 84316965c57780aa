#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab10; mkdir -p $O
B=build/lab
timeout 1200 python tools/step_lab.py --rounds=3 v2c=$B/libhns_v2c.so recplain=$B/libhns_v2d_recplain.so recnt=$B/libhns_v2d_recnt.so v2c_b=$B/libhns_v2c.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
