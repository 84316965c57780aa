#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab16; mkdir -p $O
B=build/lab
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 1200 python tools/step_lab.py --rounds=3 v3d=$B/libhns_v3d.so v3e=$B/libhns_v3e.so v3e_design1=$B/libhns_v3e.so::HNS_STEP_DESIGN=1 v3d_b=$B/libhns_v3d.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
