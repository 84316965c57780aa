#!/bin/bash
# round 3, batch 83: profiles of the reset / generator kernels at config 4's shape; predictor at the widest frames; trim time on one and two XCDs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab83; mkdir -p $O
timeout 900 bash tools/profile_envgen.sh r03_envgen > $O/prof_envgen.txt 2>&1; tail -50 $O/prof_envgen.txt
timeout 300 python tools/fps_time.py 2>&1 | grep hns_fps
timeout 600 python tools/tp_lab.py --rounds=3 --agents=6 --obst=1 --cyl=16 prod_a6c16=multi-uav-pursuit-evasion_amd/libhns.so 2>&1 | tail -3
timeout 600 python tools/tp_lab.py --rounds=3 --agents=6 --obst=1 --cyl=12 prod_a6c12=multi-uav-pursuit-evasion_amd/libhns.so 2>&1 | tail -3
timeout 600 python tools/tp_lab.py --rounds=3 --agents=3 --obst=1 --cyl=8 prod_a3c8=multi-uav-pursuit-evasion_amd/libhns.so 2>&1 | tail -3
