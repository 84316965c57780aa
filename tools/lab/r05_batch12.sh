#!/bin/bash
# round 5, batch 12: the dispatch trace again with the table taken from the middle of the run
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_b12; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-traffic-live --tp-steps 0 --stream-groups 0 --config-steps 0 --abi-steps 0"
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- $B --steps 2000 --warmup 200 > $O/trace.log 2>&1
db=$(ls $O/trace/*/*.db 2>/dev/null | head -1)
python tools/launch_overlap.py "$db" hns_step_v4_kernelILi3ELi1 256 > $O/launch_overlap.txt; head -12 $O/launch_overlap.txt
rm -rf $O/trace
