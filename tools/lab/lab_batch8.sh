#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab8; mkdir -p $O
B=build/lab
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 1200 python tools/step_lab.py --rounds=3 r1=$B/libhns_r1.so noslp=$B/libhns_v2b_noslp.so v2c=$B/libhns_v2c.so product= > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
