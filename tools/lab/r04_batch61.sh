#!/bin/bash
# round 4, batch 61: small mapping, the owners' controller-state stores as write-through stores
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b61; mkdir -p $O
timeout 600 python tools/ab_env.py HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_smsc1.so HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_smsc1.so 4096 16384 32768 --steps=3000 --blocks=5 2>&1 | grep "E=" | sed 's/ us per step.*//' | tee $O/ab.txt
