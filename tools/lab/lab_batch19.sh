#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab19; mkdir -p $O
B=build/lab
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 1200 python tools/step_lab.py --rounds=3 v3f=$B/libhns_v3f.so v3g=$B/libhns_v3g.so v3g_design1=$B/libhns_v3g.so::HNS_STEP_DESIGN=1 v3f_b=$B/libhns_v3f.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
