#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab12; mkdir -p $O
B=build/lab
timeout 1200 python tools/step_lab.py --rounds=3 v3a=$B/libhns_v3a.so prio1=$B/libhns_v3b_prio1.so prio2=$B/libhns_v3b_prio2.so prio3=$B/libhns_v3b_prio3.so v3a_b=$B/libhns_v3a.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
HNS_LIBRARY=$PWD/$B/libhns_v3b_prio3.so python tools/phase_timeline.py > $O/tl.txt 2>&1; cat $O/tl.txt
