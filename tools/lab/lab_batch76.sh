#!/bin/bash
# round 3, batch 76: torchrl-branch tests, stream-ordered setters (graph replays), manifest, bench contract
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_torchrl_branch.py tests/test_hip_parity.py tests/test_manifest.py tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -30
