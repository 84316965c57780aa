#!/bin/bash
# round 4, batch 63: the history trim at the generator's steady size (70 536 points, 36 coordinates): 2 against 4 XCDs, exchange batch 8 against others
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b63; mkdir -p $O
for x in "" 1 2 4 8; do echo "== HNS_FPS_XCDS=${x:-policy}"; if [ -n "$x" ]; then export HNS_FPS_XCDS=$x; else unset HNS_FPS_XCDS; fi; timeout 200 python tools/fps_time.py 2>&1 | grep "n=70536\|n=65536\|n=51000"; done 2>&1 | tee $O/fps.txt
