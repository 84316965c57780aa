#!/bin/bash
# round 4, batch 52: write-through (sc1) against plain stores for the outputs, re-measured on the final kernels with alternating blocks
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b52; mkdir -p $O
{ timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_stplain.so 65536
  timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_stplain.so 4096 --cylinders=5 --steps=5000
  timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_stplain.so 65536 --agents=6 --targets=2 --cylinders=16 --steps=1000 --blocks=5
  timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_stplain.so 1048576 --steps=150 --blocks=5; } 2>&1 | grep "E=" | tee $O/ab.txt
