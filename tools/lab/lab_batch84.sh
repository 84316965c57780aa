#!/bin/bash
# round 3, batch 84: predictor frames of two and three chunks: tile kernel against the weight-stationary one (HNS_TP_KERNEL)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=multi-uav-pursuit-evasion_amd/libhns.so
for k in tile ws; do
  export HNS_TP_KERNEL=$k
  timeout 600 python tools/tp_lab.py --rounds=3 --agents=6 --obst=0 ${k}_a6=$L 2>&1 | tail -1
  timeout 600 python tools/tp_lab.py --rounds=3 --agents=3 --obst=1 --cyl=5 ${k}_a3c5=$L 2>&1 | tail -1
  timeout 600 python tools/tp_lab.py --rounds=3 --agents=3 --obst=1 --cyl=8 ${k}_a3c8=$L 2>&1 | tail -1
done
