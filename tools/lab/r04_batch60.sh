#!/bin/bash
# round 4, batch 60: where the small mapping stops paying, by pursuer count (2 A + 1 waves per workgroup: residency differs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b60; mkdir -p $O
for a in 1 2 4 6 7; do for e in 8192 16384 32768; do
  timeout 300 python tools/ab_env.py HNS_STEP_MAPPING=tile HNS_STEP_MAPPING=small $e --agents=$a --steps=1500 --blocks=3 2>&1 | grep "E="; done; done | sed 's/ us per step.*//' | tee $O/ab.txt
