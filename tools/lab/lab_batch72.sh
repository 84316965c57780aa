#!/bin/bash
# round 3, batch 72: weight-stationary kernel ablations (no nonlinearities / no matrix products)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab72; mkdir -p $O
V=build/variants
timeout 1200 python tools/tp_lab.py --rounds=3 ws=$V/libhns_ws_p0.so nocell=$V/libhns_ws_nocell.so nomfma=$V/libhns_ws_nomfma.so > $O/tp_lab.txt 2>&1
cat $O/tp_lab.txt
