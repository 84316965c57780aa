#!/bin/bash
# round 4, batch 42: priorities by phase (P1 P2 P3 P4 of the pursuer waves, env wave): which tuple, and does the reverse order suit the two-round 6v2 launch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b42; mkdir -p $O
for rep in 1 2; do for lib in "" $(ls build/variants/*.so); do
  echo "== ${lib:-product}"; HNS_LIBRARY=$lib timeout 200 python tools/small_batch.py 65536 --cylinders=8 2>&1 | grep "E="
  HNS_LIBRARY=$lib timeout 200 python tools/small_batch.py 65536 --cylinders=16 --agents=6 --targets=2 --steps=1000 2>&1 | grep "E="
done; done 2>&1 | tee $O/ab.txt
