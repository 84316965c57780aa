#!/bin/bash
# round 4, batch 65: predictor kernel, wave priority by recurrence step (the workgroup of a CU that is behind goes first)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b65; mkdir -p $O
timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_wspt.so HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_wspt.so 65536 --tp --steps=600 --blocks=5 2>&1 | grep "E=" | sed 's/ us per step.*//' | tee $O/ab.txt
