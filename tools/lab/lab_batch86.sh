#!/bin/bash
# round 3, batch 86: seeded sweep over the configuration space, HIP against the oracle
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_hip_fuzz.py -q 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_hip_tp.py -q -x -k random_configuration 2>&1 | tail -40
