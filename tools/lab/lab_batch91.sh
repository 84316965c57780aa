#!/bin/bash
# round 3, batch 91: the task generator with two evaders
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_two_evaders.py tests/test_envgen.py tests/test_hip_envgen.py -q -x 2>&1 | tail -30
