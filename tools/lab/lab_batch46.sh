#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab46; mkdir -p $O
R=build/lab/libhns_v4r_lab.so
T=build/lab/libhns_v4t_lab.so
U=build/lab/libhns_v4u_lab.so
timeout 500 python tools/step_lab.py --rounds=3 v4r=$R preload=$T separgs=$U v4r_b=$R preload_b=$T separgs_b=$U > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
