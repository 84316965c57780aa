#!/bin/bash
# round 4, batch 62: ctbr / target_rate stored with the phase-2 group, write-through (instead of early and plain)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b62; mkdir -p $O
V=build/variants/libhns_ctbrlate.so
{ timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=$V HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=$V 65536
  timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=$V HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=$V 65536 --agents=6 --targets=2 --cylinders=16 --steps=1000 --blocks=5; } 2>&1 | grep "E=" | sed 's/ us per step.*//' | tee $O/ab.txt
