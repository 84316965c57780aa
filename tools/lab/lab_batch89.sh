#!/bin/bash
# round 3, batch 89: the step sweep on the first-design kernel for every shape (HNS_STEP_DESIGN=1) and on the chip-wide trim kernel (HNS_FPS_KERNEL=chip)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
HNS_STEP_DESIGN=1 timeout 1200 python -m pytest tests/test_hip_fuzz.py tests/test_two_evaders.py -q 2>&1 | tail -5
HNS_FPS_KERNEL=chip timeout 1200 python -m pytest tests/test_hip_envgen.py -q 2>&1 | tail -5
