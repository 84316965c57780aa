#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab30; mkdir -p $O
F=build/lab/libhns_v4f_lab.so
G=build/lab/libhns_v4g_lab.so
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_abi.py tests/test_two_evaders.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
HNS_LIBRARY=$F timeout 300 python tools/phase_timeline.py > $O/tl_v4f.txt 2>&1; cat $O/tl_v4f.txt
HNS_LIBRARY=$G timeout 300 python tools/phase_timeline.py > $O/tl_v4g.txt 2>&1; cat $O/tl_v4g.txt
timeout 1500 python tools/step_lab.py --rounds=3 v3=$F:0:HNS_STEP_DESIGN=3 v4f=$F v4g=$G v4f_b=$F v4g_b=$G > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
timeout 600 python tools/step_lab.py --rounds=3 --agents=6 --cyl=16 v3a6=$F:0:HNS_STEP_DESIGN=3 v4fa6=$F v4ga6=$G > $O/step_lab_a6.txt 2>&1
cat $O/step_lab_a6.txt
