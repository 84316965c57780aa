#!/bin/bash
# round 4, GPU batch 14: predictor — the activations' hi operand read once (10 ds_read_b128 per tile instead of 15); A/B against the build before
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b14
timeout 900 python -m pytest tests/test_hip_tp.py tests/test_two_evaders.py tests/test_tp_net.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6
timeout 400 python tools/tp_lab.py --rounds=5 new=multi-uav-pursuit-evasion_amd/libhns.so old=build/variants/libhns_tpold.so new2=multi-uav-pursuit-evasion_amd/libhns.so old2=build/variants/libhns_tpold.so 2>&1 | tee gpurun_out/r04b14/tp_lab.txt | tail -6
timeout 300 python tools/tp_lab.py --rounds=3 --agents=6 new=multi-uav-pursuit-evasion_amd/libhns.so old=build/variants/libhns_tpold.so 2>&1 | tee gpurun_out/r04b14/tp_lab_a6.txt | tail -4
timeout 300 python tools/tp_lab.py --rounds=3 --obst=1 new=multi-uav-pursuit-evasion_amd/libhns.so old=build/variants/libhns_tpold.so 2>&1 | tee gpurun_out/r04b14/tp_lab_obst.txt | tail -4
