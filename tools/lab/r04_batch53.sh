#!/bin/bash
# round 4, batch 53: the stores still written back at the end of the kernel (controller state; last rate, ctbr, target rate) as write-through stores too
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b53; mkdir -p $O
{ timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_sc1a.so HNS_LIBRARY=build/variants/libhns_sc1b.so 65536
  timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_sc1a.so HNS_LIBRARY=build/variants/libhns_sc1b.so 65536 --agents=6 --targets=2 --cylinders=16 --steps=1000 --blocks=5
  timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_sc1a.so HNS_LIBRARY=build/variants/libhns_sc1b.so 1048576 --steps=150 --blocks=5; } 2>&1 | grep "E=" | tee $O/ab.txt
