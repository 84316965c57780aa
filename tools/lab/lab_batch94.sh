#!/bin/bash
# round 3, batch 94: the predictor with two evaders (one launch over 2 E units): parity; the one-evader predictor and two-evader tests unchanged
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_two_evaders.py -q -x 2>&1 | tail -30
timeout 1200 python -m pytest tests/test_hip_tp.py tests/test_manifest.py -q -x 2>&1 | tail -8
for k in tile ws; do HNS_TP_KERNEL=$k timeout 600 python -m pytest tests/test_two_evaders.py -q -k predictor 2>&1 | tail -3; done
