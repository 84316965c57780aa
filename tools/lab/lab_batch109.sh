#!/bin/bash
# round 3, batch 109: sweeps of the ray-fan sensor and the Hover task
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_raycast.py tests/test_hip_hover.py -q -x 2>&1 | tail -25
