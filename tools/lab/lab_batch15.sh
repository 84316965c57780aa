#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab15; mkdir -p $O
L=build/lab/libhns_v3d_lab.so
timeout 1200 python tools/step_lab.py --rounds=3 lab0=$L:0 nostore=$L:1 noself=$L:64 nooth=$L:128 norec=$L:256 nods=$L:512 noocyl=$L:1024 nostats=$L:2048 nobig=$L:3520 lab0b=$L:0 > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
HNS_LIBRARY=$PWD/$L HNS_LAB_FLAGS=1 python tools/phase_timeline.py > $O/tl_nostore.txt 2>&1; head -3 $O/tl_nostore.txt; grep "^end" $O/tl_nostore.txt
