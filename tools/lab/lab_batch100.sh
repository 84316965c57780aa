#!/bin/bash
# round 3, batch 100: seed 381 of the deep step sweep
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HNS_FUZZ_SEEDS=3000
timeout 600 python -m pytest "tests/test_hip_fuzz.py::test_random_configuration_is_bit_exact[381]" -q -x 2>&1 | grep -v "^$" | tail -40
