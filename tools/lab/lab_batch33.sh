#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab33; mkdir -p $O
H=build/lab/libhns_v4i_lab.so
J=build/lab/libhns_v4j_lab.so
timeout 1500 python tools/step_lab.py --rounds=3 v4=$H v4_devk=$H:0:HIP_FORCE_DEV_KERNARG=1 v4_hostk=$H:0:HIP_FORCE_DEV_KERNARG=0 v4j=$J v4j_devk=$J:0:HIP_FORCE_DEV_KERNARG=1 v4_b=$H v4_devk_b=$H:0:HIP_FORCE_DEV_KERNARG=1 > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
