#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab5; mkdir -p $O
B=build/lab
timeout 1200 python tools/step_lab.py --rounds=3 r1=$B/libhns_r1.so aux16=$B/libhns_v2a_aux16.so v2b_all=$B/libhns_v2b_all.so v2b_nostats=$B/libhns_v2b_nostats.so v2b_all_nostore=$B/libhns_v2b_all.so:1 > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
HNS_LIBRARY=$PWD/$B/libhns_v2b_all.so timeout 200 python tools/phase_profile.py > $O/phase_profile.txt 2>&1; tail -8 $O/phase_profile.txt
