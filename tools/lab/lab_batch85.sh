#!/bin/bash
# round 3, batch 85: weight-stationary predictor at two chunks without the 128-register cap; three chunks by default
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=multi-uav-pursuit-evasion_amd/libhns.so
timeout 600 python tools/tp_lab.py --rounds=3 --agents=3 --obst=1 --cyl=8 default_a3c8=$L 2>&1 | tail -1
export HNS_TP_KERNEL=ws
timeout 600 python tools/tp_lab.py --rounds=3 --agents=6 --obst=0 ws_a6=$L 2>&1 | tail -1
timeout 600 python tools/tp_lab.py --rounds=3 --agents=3 --obst=1 --cyl=5 ws_a3c5=$L 2>&1 | tail -1
timeout 600 python tools/tp_lab.py --rounds=3 --agents=4 --obst=0 ws_a4=$L 2>&1 | tail -1
export HNS_TP_KERNEL=tile
timeout 600 python tools/tp_lab.py --rounds=3 --agents=4 --obst=0 tile_a4=$L 2>&1 | tail -1
