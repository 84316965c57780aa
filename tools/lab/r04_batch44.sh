#!/bin/bash
# round 4, batch 44: the priority boost as shipped (host-side rule, HNS_STEP_PRIO=0|1 override): parity, A/B by the override, shapes where the rule says no
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b44; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_reset_pid.py tests/test_hip_fuzz.py tests/test_two_evaders.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4 ) 2>&1 | tail -8
for rep in 1 2; do for pr in 0 1 ""; do echo "== HNS_STEP_PRIO=${pr:-unset}";
  HNS_STEP_PRIO=$pr timeout 200 python tools/small_batch.py 65536 --cylinders=8 2>&1 | grep "E="; done; done 2>&1 | tee $O/ab.txt
echo "== rule (unset): shapes"; timeout 200 python tools/small_batch.py 65536 --cylinders=16 --agents=6 --targets=2 --steps=1000 2>&1 | grep "E="
HNS_STEP_MAPPING=tile timeout 200 python tools/small_batch.py 49152 131072 262144 1048576 --cylinders=8 --steps=300 2>&1 | grep "E="
