#!/bin/bash
# round 4, batch 41: wave priorities against the age-ordered issue arbitration that lets the first two workgroups of a CU finish 3 us before the last two:
# 1 = by phase (laggards first), 2 = by residency slot (3 highest), 3 = slots 2-3 one step up
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b41; mkdir -p $O
for rep in 1 2; do for lib in "" build/variants/libhns_prio1.so build/variants/libhns_prio2.so build/variants/libhns_prio3.so; do
  echo "== ${lib:-product}"; HNS_LIBRARY=$lib timeout 200 python tools/small_batch.py 65536 --cylinders=8 2>&1 | grep "E="
  HNS_LIBRARY=$lib timeout 200 python tools/small_batch.py 65536 --cylinders=16 --agents=6 --targets=2 --steps=1000 2>&1 | grep "E="
done; done 2>&1 | tee $O/ab.txt
for lib in build/variants/libhns_prio1.so build/variants/libhns_prio2.so; do echo "== $lib"; HNS_LIBRARY=$lib timeout 200 python tools/phase_profile.py --envs=65536 --cylinders=8 --spread 2>&1 | grep -v amdgpu | tail -7; done | tee $O/spread.txt
