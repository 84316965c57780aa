#!/bin/bash
# round 4, batch 57: predictor kernel: window rows and observation rows as write-through stores
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b57; mkdir -p $O
{ timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_tpsc1.so HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_tpsc1.so 65536 --tp --steps=400 --blocks=5; } 2>&1 | grep "E=" | tee $O/ab.txt
timeout 600 python -m pytest tests/test_hip_tp.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
