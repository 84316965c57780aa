#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab35; mkdir -p $O
L=build/lab/libhns_v4m_lab.so
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_abi.py tests/test_two_evaders.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 1500 python tools/step_lab.py --rounds=3 v4=$L nolos1=$L:8192 v4_b=$L nolos1_b=$L:8192 > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
HNS_LAB_FLAGS=8192 HNS_LIBRARY=$L timeout 300 python tools/phase_timeline.py > $O/tl.txt 2>&1; cat $O/tl.txt
