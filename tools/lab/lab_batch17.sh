#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab17; mkdir -p $O
for e in 4096 16384 32768; do HNS_TL_ENVS=$e python tools/phase_timeline.py 2>/dev/null | grep -v "^$" > $O/tl_$e.txt; cat $O/tl_$e.txt; done
