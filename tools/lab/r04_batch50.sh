#!/bin/bash
# round 4, batch 50: small mapping, owner waves at a raised priority (they share a SIMD with their helper)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b50; mkdir -p $O
{ timeout 400 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_sop1.so HNS_LIBRARY=build/variants/libhns_sop3.so 2048 4096 16384 32768 --cylinders=5 --steps=4000 --blocks=5; } 2>&1 | grep "E=" | tee $O/ab.txt
