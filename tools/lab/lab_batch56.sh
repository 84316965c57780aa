#!/bin/bash
# round 3, batch 56: SIMD sharing microbenchmark (MFMA stream beside 1..3 VALU waves; mixed waves at 1/2/4 per SIMD; fp16 subnormals)
# and the first PMC profile of the 6-pursuer / 2-evader step kernel (config 5's shard)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab56; mkdir -p $O
timeout 200 build/mb/simd_share > $O/simd_share.txt 2>&1
cat $O/simd_share.txt
timeout 500 bash tools/profile_step.sh r03_a6t2 --agents 6 --targets 2 --cylinders 16 > $O/prof_a6t2.txt 2>&1
tail -40 $O/prof_a6t2.txt
