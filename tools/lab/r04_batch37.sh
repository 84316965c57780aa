#!/bin/bash
# round 4, batch 37: inactive flags and collision count of the k nearest cylinders carried in the sweep's keys (cylinder_pass_hits): parity; A/B of the
# tile mapping against -DHNS_NO_HITS; the small mapping with the owner's stores behind barrier 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b37; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_reset_pid.py tests/test_hip_fuzz.py tests/test_two_evaders.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -6 ) 2>&1 | tail -10
for rep in 1 2; do
for lib in "" build/variants/libhns_nohits.so; do
  for m in tile small; do echo "== lib=${lib:-product} mapping=$m"; HNS_LIBRARY=$lib HNS_STEP_MAPPING=$m timeout 200 python tools/small_batch.py 2048 4096 16384 32768 2>&1 | grep "E="; done
  echo "== lib=${lib:-product} headline / 6v2"; HNS_LIBRARY=$lib timeout 200 python tools/small_batch.py 65536 --cylinders=8 2>&1 | grep "E="
  HNS_LIBRARY=$lib timeout 200 python tools/small_batch.py 65536 --cylinders=16 --agents=6 --targets=2 --steps=1000 2>&1 | grep "E="
done; done 2>&1 | tee $O/ab.txt
timeout 200 python tools/phase_profile.py --envs=4096 --cylinders=5 --waves 2>&1 | grep -v amdgpu | tail -32 | tee $O/phase4096_small.txt
