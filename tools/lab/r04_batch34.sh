#!/bin/bash
# round 4, batch 34: first loads of a pursuer wave issued in one burst (branch-free reset_pid byte; throttle wait ahead of the ctbr stores): parity + times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b34; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_reset_pid.py tests/test_hip_fuzz.py tests/test_two_evaders.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -6 ) 2>&1 | tail -10
for rep in 1 2; do
  for m in tile small; do echo "== mapping=$m"; HNS_STEP_MAPPING=$m timeout 200 python tools/small_batch.py 2048 4096 16384 32768 2>&1 | grep "E="; done
  echo "== headline / 6v2"; timeout 200 python tools/small_batch.py 65536 --cylinders=8 2>&1 | grep "E="
  timeout 200 python tools/small_batch.py 65536 --cylinders=16 --agents=6 --targets=2 --steps=1000 2>&1 | grep "E="
done 2>&1 | tee $O/times.txt
