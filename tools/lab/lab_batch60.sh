#!/bin/bash
# round 3, batch 60: predictor ablations — no nonlinearities / no matrix products / deeper operand ring
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab60; mkdir -p $O
V=build/variants
timeout 900 python tools/tp_lab.py --rounds=3 nopk=$V/libhns_nopk.so nocell=$V/libhns_nocell.so nomfma=$V/libhns_nomfma.so depth3=$V/libhns_depth3.so depth4=$V/libhns_depth4.so nopk_b=$V/libhns_nopk.so > $O/tp_lab.txt 2>&1
cat $O/tp_lab.txt
