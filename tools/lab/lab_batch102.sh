#!/bin/bash
# round 3, batch 102: sweep over the generator's configurations (whole env)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_envgen.py -q -x -k random_configurations 2>&1 | tail -30
