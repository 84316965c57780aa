#!/bin/bash
# round 3, batch 95: the predictor sweep with one or two evaders drawn at random
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_hip_tp.py -q -x -k random_configuration 2>&1 | tail -12
