#!/bin/bash
# round 3, batch 61: cycles per phase of the predictor's timestep (full kernel, without matrix products, without nonlinearities)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab61; mkdir -p $O
V=build/variants
for v in phases phases_nomfma phases_nocell; do
  echo "== $v" >> $O/phases.txt
  HNS_LIBRARY=$V/libhns_$v.so timeout 300 python tools/tp_phase_profile.py --phases >> $O/phases.txt 2>&1
done
cat $O/phases.txt
