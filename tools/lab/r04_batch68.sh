#!/bin/bash
# round 4, batch 68: predictor prologue in one memory round trip (rows 1-2 and the frame's values issued together; the frame's four values + detection byte
# from one asm block): parity, A/B against the build before, phase stamps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b68; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_tp.py tests/test_two_evaders.py tests/test_manifest.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
timeout 600 python tools/ab_env.py HNS_LIBRARY=build/variants/libhns_before.so HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so HNS_LIBRARY=build/variants/libhns_before.so HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so 65536 --tp --steps=600 --blocks=5 2>&1 | grep "E=" | sed 's/ us per step.*//' | tee $O/ab.txt
timeout 200 python tools/tp_phases.py 65536 2>&1 | grep -v amdgpu | tail -8 | tee $O/tp_phases.txt
