#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b8
for x in 1 2 4; do echo "HNS_FPS_XCDS=$x"; HNS_FPS_XCDS=$x timeout 200 python tools/fps_time.py 2>&1 | tail -6; done | tee gpurun_out/r04b8/fps_time.txt
timeout 600 python -m pytest tests/test_hip_envgen.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
