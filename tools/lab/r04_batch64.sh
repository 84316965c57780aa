#!/bin/bash
# round 4, batch 64: the history trim on THREE XCDs (65 536 < n <= 98 304: the generator's steady 70 536 points): parity of the trim tests with 3 forced, times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b64; mkdir -p $O
HNS_FPS_XCDS=3 timeout 900 python -m pytest tests/test_hip_envgen.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
timeout 900 python -m pytest tests/test_hip_envgen.py tests/test_envgen.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
for x in "" 3 4; do echo "== HNS_FPS_XCDS=${x:-policy}"; if [ -n "$x" ]; then export HNS_FPS_XCDS=$x; else unset HNS_FPS_XCDS; fi; timeout 200 python tools/fps_time.py 2>&1 | grep "n=70536\|n=65536\|n=69632"; done 2>&1 | tee $O/fps.txt
