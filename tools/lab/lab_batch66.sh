#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab66; mkdir -p $O
HNS_LIBRARY=build/variants/libhns_ws.so timeout 300 python tools/tp_debug.py --none > $O/dbg.txt 2>&1
grep -v amdgpu $O/dbg.txt | cut -c1-330
