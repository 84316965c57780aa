#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab42; mkdir -p $O
Q=build/lab/libhns_v4q_lab.so
R=build/lab/libhns_v4r_lab.so
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_abi.py tests/test_two_evaders.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 400 python tools/step_lab.py --rounds=3 v4q=$Q v4r=$R v4q_b=$Q v4r_b=$R > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
timeout 300 python tools/step_lab.py --rounds=3 --agents=6 --cyl=16 --targets=2 v1_a6t2=$R:0:HNS_STEP_DESIGN=1 v4_a6t2=$R > $O/step_lab_a6.txt 2>&1
cat $O/step_lab_a6.txt
