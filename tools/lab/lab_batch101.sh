#!/bin/bash
# round 3, batch 101: the step sweep at 3000 seeds again (the oracle follows the env's evader-speed curriculum now)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HNS_FUZZ_SEEDS=3000
timeout 1500 python -m pytest tests/test_hip_fuzz.py -q -x -k "bit_exact" 2>&1 | tail -8
HNS_STEP_DESIGN=1 timeout 1500 python -m pytest tests/test_hip_fuzz.py -q -x -k "bit_exact" 2>&1 | tail -4
