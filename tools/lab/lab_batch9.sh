#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab9; mkdir -p $O
export HNS_LIBRARY=$PWD/build/lab/libhns_v2c.so
python tools/phase_timeline.py > $O/tl_base.txt 2>&1; cat $O/tl_base.txt
HNS_LAB_STAGGER=4 python tools/phase_timeline.py > $O/tl_stag4.txt 2>&1; cat $O/tl_stag4.txt
HNS_LAB_FLAGS=1 python tools/phase_timeline.py > $O/tl_nostore.txt 2>&1; cat $O/tl_nostore.txt
