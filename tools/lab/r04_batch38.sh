#!/bin/bash
# round 4, batch 38: small mapping, which of the last two changes costs 0.8 us: x = hits in keys + stores behind b3, y = hits + stores before b3, z = neither (batch 34's kernel), w = stores behind b3 only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b38; mkdir -p $O
for rep in 1 2; do for n in x y z w; do echo "== $n"; HNS_LIBRARY=build/variants/libhns_sm_$n.so timeout 200 python tools/small_batch.py 2048 4096 16384 2>&1 | grep "E="; done; done 2>&1 | tee $O/ab.txt
