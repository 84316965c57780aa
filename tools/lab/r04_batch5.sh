#!/bin/bash
# round 4, GPU batch 5: farthest-point trim with the candidates' rows through LDS; env.rollout / learner consumers (stand-in torchrl)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b5
( timeout 900 python -m pytest tests/test_hip_envgen.py tests/test_torchrl_branch.py tests/test_manifest.py tests/test_host_logic.py -m gpu -q -p no:cacheprovider ) > gpurun_out/r04b5/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/r04b5/pytest.log | cut -c1-260 | head -40
for b in 1 4; do echo "HNS_FPS_BATCH=$b"; HNS_FPS_BATCH=$b timeout 200 python tools/fps_time.py 2>&1 | tail -6; done | tee gpurun_out/r04b5/fps_time.txt
