#!/bin/bash
# round 4, batch 54: write-through for the controller-state stores (level 1), + last rate (3), + progress (4)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b54; mkdir -p $O
L="HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_sc1l1.so HNS_LIBRARY=build/variants/libhns_sc1l3.so HNS_LIBRARY=build/variants/libhns_sc1l4.so"
{ timeout 600 python tools/ab_env.py $L 65536
  timeout 600 python tools/ab_env.py $L 65536 --agents=6 --targets=2 --cylinders=16 --steps=1000 --blocks=5
  timeout 600 python tools/ab_env.py $L 262144 --steps=500 --blocks=5
  HNS_STEP_MAPPING=tile timeout 600 python tools/ab_env.py $L 32768 --steps=4000 --blocks=5; } 2>&1 | grep "E=" | tee $O/ab.txt
