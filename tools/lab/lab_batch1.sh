#!/bin/bash
# batch 1: instruction rates, copy rates, ablations of the round-1 step kernel
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/lab1
O=gpurun_out/lab1
timeout 120 build/lab/valu_rate > $O/valu_rate.txt 2>&1
timeout 200 python tools/copy_rate.py > $O/copy_rate.txt 2>&1
L=build/lab/libhns_lab.so
timeout 900 python tools/step_lab.py base= lab0=$L:0 nostore=$L:1 nop1=$L:2 nop2=$L:4 nop3a=$L:8 nop3b=$L:16 nocompute=$L:30 loadsonly=$L:31 storesonly=$L:62 \
    empty=$L:63 noload=$L:32 fastdiv=build/lab/libhns_fastdiv.so base2= > $O/step_lab.txt 2>&1
timeout 200 python tools/phase_profile.py > $O/phase_profile.txt 2>&1
cat $O/valu_rate.txt $O/copy_rate.txt $O/step_lab.txt $O/phase_profile.txt
