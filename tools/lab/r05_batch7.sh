#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_b7; mkdir -p $O
timeout 600 python -m pytest tests/test_envgen.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
EP_LEN=800 timeout 300 python tools/envgen_cost.py 2>&1 | grep -v amdgpu.ids | tee $O/envgen_cost.txt
