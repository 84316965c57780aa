#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab2; mkdir -p $O
timeout 200 build/lab/valu_rate > $O/valu_rate.txt 2>&1
L=build/lab/libhns_lab.so
timeout 1200 python tools/step_lab.py --rounds=3 lab0=$L:0 noself=$L:64 nooth=$L:128 norec=$L:256 nods=$L:512 noocyl=$L:1024 nostats=$L:2048 \
    stag1=$L:0:HNS_LAB_STAGGER=1 stag2=$L:0:HNS_LAB_STAGGER=2 stag4=$L:0:HNS_LAB_STAGGER=4 stag6=$L:0:HNS_LAB_STAGGER=6 stag8=$L:0:HNS_LAB_STAGGER=8 \
    nost_stag4=$L:1:HNS_LAB_STAGGER=4 lab0b=$L:0 > $O/step_lab.txt 2>&1
cat $O/valu_rate.txt $O/step_lab.txt
