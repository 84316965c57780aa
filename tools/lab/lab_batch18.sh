#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab18; mkdir -p $O
B=build/lab
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 1200 python tools/step_lab.py --rounds=3 v3f=$B/libhns_v3f.so v3f_design1=$B/libhns_v3f.so::HNS_STEP_DESIGN=1 v3f_b=$B/libhns_v3f.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
python tools/phase_timeline.py > $O/tl.txt 2>&1; cat $O/tl.txt | grep -v "obs done  *-"
HNS_TL_ENVS=16384 python tools/phase_timeline.py 2>&1 | grep -v "obs done  *-" > $O/tl16k.txt; cat $O/tl16k.txt
