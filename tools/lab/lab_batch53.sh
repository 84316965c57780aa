#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab53; mkdir -p $O
for tag in chip xcd chip2 xcd2; do
  if [[ $tag == chip* ]]; then export HNS_FPS_KERNEL=chip; else unset HNS_FPS_KERNEL; fi
  EP_LEN=60 timeout 200 python tools/envgen_cost.py > $O/$tag.txt 2>&1
  echo "== $tag"; grep "episode\|total" $O/$tag.txt | cut -c1-110
done
