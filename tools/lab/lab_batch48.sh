#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab48; mkdir -p $O
B=build/lab
timeout 600 python tools/step_lab.py --rounds=3 final=$B/libhns_final_lab.so stplain=$B/libhns_stplain_lab.so outplain=$B/libhns_outplain_lab.so allplain=$B/libhns_allplain_lab.so final_b=$B/libhns_final_lab.so stplain_b=$B/libhns_stplain_lab.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
