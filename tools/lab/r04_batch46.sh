#!/bin/bash
# round 4, batch 46: priority boost, alternating blocks inside one process (tools/ab_env.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b46; mkdir -p $O
{ HNS_STEP_MAPPING=tile timeout 600 python tools/ab_env.py HNS_STEP_PRIO=0 HNS_STEP_PRIO=1 49152 65536 131072 262144 --steps=1500 --blocks=5
  timeout 300 python tools/ab_env.py HNS_STEP_PRIO=0 HNS_STEP_PRIO=1 65536 --cylinders=5
  timeout 300 python tools/ab_env.py HNS_STEP_PRIO=0 HNS_STEP_PRIO=1 65536 --agents=2
  timeout 300 python tools/ab_env.py HNS_STEP_PRIO=0 HNS_STEP_PRIO=1 65536 --agents=4 --steps=2000
  timeout 300 python tools/ab_env.py HNS_STEP_PRIO=0 HNS_STEP_PRIO=1 65536 --agents=6 --targets=2 --cylinders=16 --steps=1000 --blocks=5; } 2>&1 | grep "E=" | tee $O/ab.txt
