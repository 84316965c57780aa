#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab26; mkdir -p $O
L=build/lab/libhns_v4_lab.so
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_abi.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
HNS_LIBRARY=$L timeout 300 python tools/wave_placement.py > $O/placement.txt 2>&1; tail -32 $O/placement.txt
HNS_LIBRARY=$L timeout 300 python tools/phase_timeline.py > $O/tl_v4.txt 2>&1; cat $O/tl_v4.txt
timeout 1500 python tools/step_lab.py --rounds=3 v3=$L:0:HNS_STEP_DESIGN=3 v4=$L v4r1=$L:0:HNS_LAB_STAGGER=1 v4r4=$L:0:HNS_LAB_STAGGER=4 v4r6=$L:0:HNS_LAB_STAGGER=6 v4r9=$L:0:HNS_LAB_STAGGER=9 v3r9=$L:0:HNS_STEP_DESIGN=3,HNS_LAB_STAGGER=9 v4_b=$L > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
