#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab3; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
L=build/lab/libhns_v2a.so
timeout 1200 python tools/step_lab.py --rounds=3 r1=build/lab/libhns_r1.so v2a=$L:0 v2a_nostore=$L:1 v2a_noself=$L:64 v2a_nooth=$L:128 v2a_norec=$L:256 v2a_noocyl=$L:1024 v2a_nocompute=$L:30 r1b=build/lab/libhns_r1.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
timeout 200 python tools/phase_profile.py > $O/phase_profile.txt 2>&1; cat $O/phase_profile.txt
