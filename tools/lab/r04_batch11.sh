#!/bin/bash
# round 4, GPU batch 11: farthest-point trim with eight candidates per exchange
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b11
timeout 600 python -m pytest tests/test_hip_envgen.py tests/test_envgen.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for b in 4 8; do echo "HNS_FPS_BATCH=$b"; HNS_FPS_BATCH=$b timeout 200 python tools/fps_time.py 2>&1 | tail -6; done | tee gpurun_out/r04b11/fps_time.txt
echo "HNS_FPS_XCDS=2 (batch 8)"; HNS_FPS_XCDS=2 timeout 200 python tools/fps_time.py 2>&1 | tail -6 | tee -a gpurun_out/r04b11/fps_time.txt
echo "HNS_FPS_XCDS=4 (batch 8)"; HNS_FPS_XCDS=4 timeout 200 python tools/fps_time.py 2>&1 | tail -6 | tee -a gpurun_out/r04b11/fps_time.txt
