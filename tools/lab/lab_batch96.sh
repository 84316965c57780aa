#!/bin/bash
# round 3, batch 96: whole GPU suite + smoke() on the build with the two-evader predictor / generator and the tightened golden tolerances
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
