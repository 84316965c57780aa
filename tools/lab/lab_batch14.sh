#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab14; mkdir -p $O
B=build/lab
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 1200 python tools/step_lab.py --rounds=3 v3c=$B/libhns_v3c.so v3d=$B/libhns_v3d.so v3c_b=$B/libhns_v3c.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
HNS_LIBRARY=$PWD/$B/libhns_v3d.so python tools/phase_timeline.py > $O/tl.txt 2>&1; cat $O/tl.txt
