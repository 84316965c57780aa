#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab34; mkdir -p $O
H=build/lab/libhns_v4i_lab.so
L=build/lab/libhns_v4l_lab.so
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_abi.py tests/test_two_evaders.py tests/test_env_api.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 1500 python tools/step_lab.py --rounds=3 v4i=$H v4l=$L v4i_b=$H v4l_b=$L > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
HNS_LIBRARY=$L timeout 300 python tools/phase_timeline.py > $O/tl.txt 2>&1; cat $O/tl.txt
timeout 600 python tools/step_lab.py --rounds=3 --agents=6 --cyl=16 v4ia6=$H v4la6=$L > $O/step_lab_a6.txt 2>&1
cat $O/step_lab_a6.txt
