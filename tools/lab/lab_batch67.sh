#!/bin/bash
# round 3, batch 67: per-hidden-unit deviation of the weight-stationary kernel: default build, 2 waves per SIMD (256 registers), wait states before the cell update
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab67; mkdir -p $O
for v in ws ws_occ2 ws_nops; do echo "== $v" >> $O/dbg2.txt; HNS_LIBRARY=build/variants/libhns_$v.so timeout 300 python tools/tp_debug2.py >> $O/dbg2.txt 2>&1; done
echo "== tile kernel" >> $O/dbg2.txt; HNS_TP_KERNEL=tile HNS_LIBRARY=build/variants/libhns_ws.so timeout 300 python tools/tp_debug2.py >> $O/dbg2.txt 2>&1
grep -v amdgpu $O/dbg2.txt | cut -c1-200
