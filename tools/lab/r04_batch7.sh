#!/bin/bash
# round 4, GPU batch 7: farthest-point trim — XCDs at work (1 / 2 / 4 / 8) with four samples per exchange
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b7
for x in 1 2 4 8; do echo "HNS_FPS_XCDS=$x"; HNS_FPS_XCDS=$x timeout 200 python tools/fps_time.py 2>&1 | tail -6; done | tee gpurun_out/r04b7/fps_time.txt
HNS_FPS_XCDS=4 timeout 600 python -m pytest tests/test_hip_envgen.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
HNS_FPS_XCDS=8 timeout 600 python -m pytest tests/test_hip_envgen.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
