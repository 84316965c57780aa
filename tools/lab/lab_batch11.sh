#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab11; mkdir -p $O
B=build/lab
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 1200 python tools/step_lab.py --rounds=3 r1=$B/libhns_r1.so v2c=$B/libhns_v2c.so v3a=$B/libhns_v3a.so v3a_design1=$B/libhns_v3a.so::HNS_STEP_DESIGN=1 > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
HNS_LIBRARY=$PWD/$B/libhns_v3a.so python tools/phase_timeline.py > $O/tl_v3a.txt 2>&1; cat $O/tl_v3a.txt
