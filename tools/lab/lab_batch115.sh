#!/bin/bash
# round 3, batch 115: bench contract + sharding tests after the lazy success rate
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_bench_contract.py tests/test_sharding_gloo.py -q -x 2>&1 | tail -3
