#!/bin/bash
# round 4, batch 59: step + predictor as two half batches on two streams inside env.step (task.tp_overlap), alternating blocks in one process
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b59; mkdir -p $O
timeout 600 python tools/ab_env.py task.tp_overlap=0 task.tp_overlap=1 task.tp_overlap=0 task.tp_overlap=1 65536 --tp --steps=600 --blocks=5 2>&1 | grep "E=" | tee $O/ab.txt
