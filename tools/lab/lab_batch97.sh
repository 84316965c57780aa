#!/bin/bash
# round 3, batch 97: weight-stationary predictor at two chunks under the 128-register cap with shorter operand rings (scratch 144 / 128 / 40 B) against the defaults
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HNS_TP_KERNEL=ws
for v in occ4_d3 occ4_d2 occ4_d1; do
  timeout 600 python tools/tp_lab.py --rounds=3 --agents=6 --obst=0 ws2_$v=build/variants/libhns_ws2_$v.so 2>&1 | tail -1
done
timeout 600 python tools/tp_lab.py --rounds=3 --agents=6 --obst=0 ws2_default=multi-uav-pursuit-evasion_amd/libhns.so 2>&1 | tail -1
HNS_TP_KERNEL=tile timeout 600 python tools/tp_lab.py --rounds=3 --agents=6 --obst=0 tile2_default=multi-uav-pursuit-evasion_amd/libhns.so 2>&1 | tail -1
