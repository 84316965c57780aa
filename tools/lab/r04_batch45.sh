#!/bin/bash
# round 4, batch 45: priority boost, same-box A/B by the override at the sizes around the rule's limits; the rule itself with the variable unset
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b45; mkdir -p $O
for rep in 1 2; do for pr in 0 1; do echo "== HNS_STEP_PRIO=$pr";
  HNS_STEP_PRIO=$pr HNS_STEP_MAPPING=tile timeout 300 python tools/small_batch.py 32768 49152 65536 98304 131072 196608 262144 --cylinders=8 --steps=400 2>&1 | grep "E="; done; done 2>&1 | tee $O/ab.txt
echo "== unset"; timeout 300 python tools/small_batch.py 65536 131072 262144 --cylinders=8 --steps=400 2>&1 | grep "E="
