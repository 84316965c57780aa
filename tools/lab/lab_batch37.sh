#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab37; mkdir -p $O
N=build/lab/libhns_v4n_lab.so
L=build/lab/libhns_v4o_lab.so
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_abi.py tests/test_two_evaders.py tests/test_hip_envgen.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 400 python tools/step_lab.py --rounds=3 v4n=$N v4o=$L v4n_b=$N v4o_b=$L > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
HNS_LIBRARY=$L timeout 200 python tools/phase_timeline.py > $O/tl.txt 2>&1; cat $O/tl.txt
timeout 300 python tools/step_lab.py --rounds=3 --agents=6 --cyl=16 v4na6=$N v4oa6=$L > $O/step_lab_a6.txt 2>&1
cat $O/step_lab_a6.txt
