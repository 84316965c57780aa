#!/bin/bash
# round 4, batch 25: warming the scalar cache with the parameter block at wave start: A/B (product vs -DHNS_NO_WARM), both mappings, and the headline shape
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b25; mkdir -p $O
for rep in 1 2; do
for lib in "" build/variants/libhns_nowarm.so; do for m in tile small; do echo "== lib=${lib:-product} mapping=$m"; HNS_LIBRARY=$lib HNS_STEP_MAPPING=$m timeout 200 python tools/small_batch.py 2048 4096 16384 65536 2>&1 | grep "E="; done; done
done 2>&1 | tee $O/ab.txt
timeout 200 python tools/phase_profile.py --envs=4096 --cylinders=5 --mapping=small --waves 2>&1 | grep -v amdgpu | tail -32 | tee $O/phase4096_small.txt
