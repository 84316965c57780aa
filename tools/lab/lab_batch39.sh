#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab39; mkdir -p $O
N=build/lab/libhns_v4p_lab.so
L=build/lab/libhns_v4q_lab.so
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_abi.py tests/test_two_evaders.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 400 python tools/step_lab.py --rounds=3 v4p=$N v4q=$L v4p_b=$N v4q_b=$L > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
HNS_LIBRARY=$L timeout 200 python tools/phase_timeline.py > $O/tl.txt 2>&1; cat $O/tl.txt
