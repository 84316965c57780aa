#!/bin/bash
# round 4, batch 66: predictor priority by recurrence step: threshold variants (1: t*4/T; 2: two levels; 3: 2-3-3-2 steps; 4: 1-2-3-4 steps)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b66; mkdir -p $O
timeout 900 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_wspt.so HNS_LIBRARY=build/variants/libhns_wspt2.so HNS_LIBRARY=build/variants/libhns_wspt3.so HNS_LIBRARY=build/variants/libhns_wspt4.so 65536 --tp --steps=600 --blocks=5 2>&1 | grep "E=" | sed 's/ us per step.*//' | tee $O/ab.txt
