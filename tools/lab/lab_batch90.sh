#!/bin/bash
# round 3, batch 90: snapshot / resume over random configurations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_hip_fuzz.py -q -x -k snapshot 2>&1 | tail -30
