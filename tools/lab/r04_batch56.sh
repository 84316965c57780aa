#!/bin/bash
# round 4, batch 56: write-through controller-state stores past the Infinity Cache, both orders of the two settings (the first env of a process runs ~3 % faster at 262 144 envs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b56; mkdir -p $O
V=build/variants/libhns_sc1l1.so
{ timeout 600 python tools/ab_env.py HNS_LIBRARY=$V HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=$V HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so 262144 --steps=500 --blocks=5
  timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=$V HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=$V 262144 --steps=500 --blocks=5
  timeout 600 python tools/ab_env.py HNS_LIBRARY=$V HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=$V 1048576 --steps=150 --blocks=5
  timeout 600 python tools/ab_env.py HNS_LIBRARY=$V HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=$V HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so 131072 --steps=1000 --blocks=5; } 2>&1 | grep "E=" | tee $O/ab.txt
