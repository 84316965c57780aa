#!/bin/bash
# round 3, batch 73: weight-stationary kernel ablations — no frame phase / no barriers / both / skeleton (neither products nor cell update)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab73; mkdir -p $O
V=build/variants
timeout 1200 python tools/tp_lab.py --rounds=3 ws=$V/libhns_ws_p0.so noframe=$V/libhns_ws_a1.so nobarrier=$V/libhns_ws_a2.so neither=$V/libhns_ws_a3.so skeleton=$V/libhns_ws_skel.so 2>&1 | cut -c1-100 > $O/tp_lab.txt
cat $O/tp_lab.txt
