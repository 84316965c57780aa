#!/bin/bash
# round 4, batch 71: farthest-point trim, up to 16 candidates examined per exchange (each workgroup still publishes its top 8): parity of the trim and generator tests, times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b71; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_envgen.py tests/test_envgen.py tests/test_two_evaders.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
HNS_FUZZ_GEN_SEEDS=200 timeout 900 python -m pytest tests/test_hip_envgen.py -x -q -m gpu -k random -p no:cacheprovider 2>&1 | tail -2
timeout 200 python tools/fps_time.py 2>&1 | grep "hns_fps" | tee $O/fps.txt
HNS_FPS_BATCH=4 timeout 200 python tools/fps_time.py 2>&1 | grep "n=70536"
