#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab31; mkdir -p $O
H=build/lab/libhns_v4h_lab.so
timeout 1500 python tools/step_lab.py --rounds=3 v4=$H p_r=$H:0:HNS_LAB_STAGGER=1 p_inv=$H:0:HNS_LAB_STAGGER=2 p_r_env3=$H:0:HNS_LAB_STAGGER=3 v4_b=$H p_r_b=$H:0:HNS_LAB_STAGGER=1 > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
HNS_LAB_STAGGER=1 HNS_LIBRARY=$H timeout 300 python tools/phase_timeline.py > $O/tl_p_r.txt 2>&1; cat $O/tl_p_r.txt
