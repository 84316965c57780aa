#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab51; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_envgen.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
echo "--- XCD-local where it applies"; timeout 120 python tools/fps_time.py 2>&1 | tail -4
echo "--- chip-wide kernel"; HNS_FPS_KERNEL=chip timeout 120 python tools/fps_time.py 2>&1 | tail -4
