#!/bin/bash
# round 4, batch 55: how much two envs of the SAME build differ inside one process (buffer placement), i.e. the noise floor of tools/ab_env.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b55; mkdir -p $O
{ timeout 600 python tools/ab_env.py HNS_DUMMY=1 HNS_DUMMY=2 HNS_DUMMY=3 HNS_DUMMY=4 65536
  timeout 600 python tools/ab_env.py HNS_DUMMY=1 HNS_DUMMY=2 HNS_DUMMY=3 HNS_DUMMY=4 262144 --steps=500 --blocks=5
  timeout 600 python tools/ab_env.py HNS_DUMMY=1 HNS_DUMMY=2 HNS_DUMMY=3 1048576 --steps=150 --blocks=5; } 2>&1 | grep "E=" | tee $O/ab.txt
