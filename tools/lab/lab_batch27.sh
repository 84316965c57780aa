#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab27; mkdir -p $O
L=build/lab/libhns_v4c_lab.so
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_abi.py tests/test_two_evaders.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
HNS_LIBRARY=$L timeout 300 python tools/phase_timeline.py > $O/tl_v4.txt 2>&1; cat $O/tl_v4.txt
HNS_STEP_DESIGN=3 HNS_LIBRARY=$L timeout 300 python tools/phase_timeline.py > $O/tl_v3.txt 2>&1; cat $O/tl_v3.txt
timeout 1500 python tools/step_lab.py --rounds=3 v3=$L:0:HNS_STEP_DESIGN=3 v4=$L v3_b=$L:0:HNS_STEP_DESIGN=3 v4_b=$L > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
timeout 600 python tools/step_lab.py --rounds=3 --agents=6 --cyl=16 v3a6=$L:0:HNS_STEP_DESIGN=3 v4a6=$L > $O/step_lab_a6.txt 2>&1
cat $O/step_lab_a6.txt
