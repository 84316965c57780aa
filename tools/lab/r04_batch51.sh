#!/bin/bash
# round 4, batch 51: predictor kernel, wave priorities around the matrix burst / the cell update (does the SIMD overlap one wave's MFMAs with another's vector work when told whom to prefer?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b51; mkdir -p $O
{ timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_wsp10.so HNS_LIBRARY=build/variants/libhns_wsp01.so HNS_LIBRARY=build/variants/libhns_wsp30.so 65536 --tp --steps=400 --blocks=5; } 2>&1 | grep "E=" | tee $O/ab.txt
