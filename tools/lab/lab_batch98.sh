#!/bin/bash
# round 3, batch 98: whole episodes at the full batch size, bit for bit
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 1200 python -m pytest tests/test_hip_parity.py -q -x -k full_size_episodes 2>&1 | tail -12 ) 2>&1 | tail -16
