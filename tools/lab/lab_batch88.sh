#!/bin/bash
# round 3, batch 88: generator sweeps (trim shapes, perturbation + task reset), predictor sweep with masked resets
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_hip_envgen.py -q -x -k "random" 2>&1 | tail -30
timeout 1200 python -m pytest tests/test_hip_tp.py -q -x -k "random_configuration" 2>&1 | tail -30
