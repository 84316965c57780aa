#!/bin/bash
# round 3, batch 87: the predictor sweep with each kernel forced (tile: 1-3 chunks, ws: all)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for k in tile ws; do
  HNS_TP_KERNEL=$k timeout 1200 python -m pytest tests/test_hip_tp.py -q -k "random_configuration or matches_oracle or golden" 2>&1 | tail -4
done
