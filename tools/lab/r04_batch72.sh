#!/bin/bash
# round 4, batch 72: past two residency rounds: finishers first (priority 1 from phase 3a on) instead of laggards first — 262 144 and 1 048 576 envs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b72; mkdir -p $O
{ HNS_LIBRARY=build/variants/libhns_prev.so timeout 600 python tools/ab_env.py HNS_STEP_PRIO=0 HNS_STEP_PRIO=1 HNS_STEP_PRIO=0 HNS_STEP_PRIO=1 262144 --steps=500 --blocks=5
  HNS_LIBRARY=build/variants/libhns_prev.so timeout 600 python tools/ab_env.py HNS_STEP_PRIO=0 HNS_STEP_PRIO=1 HNS_STEP_PRIO=0 1048576 --steps=150 --blocks=5
  HNS_LIBRARY=build/variants/libhns_prev.so timeout 600 python tools/ab_env.py HNS_STEP_PRIO=0 HNS_STEP_PRIO=1 HNS_STEP_PRIO=0 HNS_STEP_PRIO=1 65536 --agents=6 --targets=2 --cylinders=16 --steps=1000 --blocks=5; } 2>&1 | grep "E=" | sed 's/ us per step.*//' | tee $O/ab.txt
