#!/bin/bash
# round 3, batch 59: predictor with packed fp32 (round 2) against plain fp32 instructions
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab59; mkdir -p $O
timeout 600 python tools/tp_lab.py --rounds=5 pk=build/variants/libhns_pk.so nopk=build/variants/libhns_nopk.so pk_b=build/variants/libhns_pk.so nopk_b=build/variants/libhns_nopk.so > $O/tp_lab.txt 2>&1
cat $O/tp_lab.txt
