#!/bin/bash
# round 3, batch 104: examples/rollout.py (with and without the predictor) still runs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python examples/rollout.py --envs 16384 --rollouts 3 2>&1 | grep -v amdgpu | tail -6
timeout 300 python examples/rollout.py --envs 16384 --rollouts 3 --tp 2>&1 | grep -v amdgpu | tail -6
