#!/bin/bash
# round 3, batch 64: where the weight-stationary kernel deviates from the oracle (parameter ablations, repeatability, two formula variants)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab64; mkdir -p $O
for v in ws_scaled; do echo "== $v" >> $O/dbg.txt; HNS_LIBRARY=build/variants/libhns_$v.so timeout 300 python tools/tp_debug.py >> $O/dbg.txt 2>&1; done
cat $O/dbg.txt
