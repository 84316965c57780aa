#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab44; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "line_of_sight" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 400 python tools/soak.py 1500 > $O/soak.txt 2>&1; tail -6 $O/soak.txt
