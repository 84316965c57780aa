#!/bin/bash
# round 4, batch 47: tile mapping at 65 536 envs / 6v2, alternating blocks in one process: kernel-argument preload, max-ilp scheduling
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b47; mkdir -p $O
{ timeout 400 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_tilepre.so HNS_LIBRARY=build/variants/libhns_maxilp.so 65536
  timeout 400 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_tilepre.so HNS_LIBRARY=build/variants/libhns_maxilp.so 65536 --agents=6 --targets=2 --cylinders=16 --steps=1000 --blocks=5
  timeout 400 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_tilepre.so HNS_LIBRARY=build/variants/libhns_maxilp.so 4096 --cylinders=5 --steps=5000; } 2>&1 | grep "E=" | tee $O/ab.txt
