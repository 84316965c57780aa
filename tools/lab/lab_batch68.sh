#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab68; mkdir -p $O
HNS_LIBRARY=build/variants/libhns_ws_dbg.so timeout 300 python tools/tp_debug3.py 459 51 > $O/dbg3.txt 2>&1
grep -v amdgpu $O/dbg3.txt | cut -c1-400
