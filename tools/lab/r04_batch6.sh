#!/bin/bash
# round 4, GPU batch 6: farthest-point trim — wave maxima on the DPP ladder, one generic instantiation
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b6
( timeout 900 python -m pytest tests/test_hip_envgen.py tests/test_hip_tp.py -m gpu -q -p no:cacheprovider ) > gpurun_out/r04b6/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/r04b6/pytest.log | cut -c1-260 | head -40
for b in 1 2 4; do echo "HNS_FPS_BATCH=$b"; HNS_FPS_BATCH=$b timeout 200 python tools/fps_time.py 2>&1 | tail -6; done | tee gpurun_out/r04b6/fps_time.txt
