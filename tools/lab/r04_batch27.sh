#!/bin/bash
# round 4, batch 27: scheduling strategy A/B (-mllvm -amdgpu-sched-strategy=max-ilp): small batches in both mappings, the headline shape, the 6v2 shard
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b27; mkdir -p $O
for rep in 1 2; do
for lib in "" build/variants/libhns_maxilp.so; do
  for m in tile small; do echo "== lib=${lib:-product} mapping=$m"; HNS_LIBRARY=$lib HNS_STEP_MAPPING=$m timeout 200 python tools/small_batch.py 2048 4096 16384 2>&1 | grep "E="; done
  echo "== lib=${lib:-product} headline / 6v2"; HNS_LIBRARY=$lib timeout 200 python tools/small_batch.py 65536 --cylinders=8 2>&1 | grep "E="
  HNS_LIBRARY=$lib timeout 200 python tools/small_batch.py 65536 --cylinders=16 --agents=6 --targets=2 --steps=1000 2>&1 | grep "E="
done; done 2>&1 | tee $O/ab.txt
