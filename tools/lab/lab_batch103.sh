#!/bin/bash
# round 3, batch 103: the step sweep with env index offsets (shards of a larger batch)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
HNS_FUZZ_SEEDS=1000 timeout 1500 python -m pytest tests/test_hip_fuzz.py -q -x -k "bit_exact" 2>&1 | tail -5
