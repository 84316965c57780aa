#!/bin/bash
# round 3, batch 58: SIMD sharing microbenchmark v2 (per-instruction-kind VALU streams beside an MFMA stream; mixed waves with / without packed fp32)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab58; mkdir -p $O
timeout 300 build/mb/simd_share > $O/simd_share.txt 2>&1
cat $O/simd_share.txt
