#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab32; mkdir -p $O
H=build/lab/libhns_v4i_lab.so
HNS_LIBRARY=$H timeout 300 python tools/phase_timeline.py > $O/tl.txt 2>&1; cat $O/tl.txt
timeout 1500 python tools/step_lab.py --rounds=3 v4=$H nolos1=$H:8192 nolos2=$H:16384 nolos12=$H:24576 noself=$H:64 nooth=$H:128 noocyl=$H:1024 nods=$H:512 norec=$H:256 nostats=$H:2048 nostore=$H:1 noobs=$H:1216 v4_b=$H > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
