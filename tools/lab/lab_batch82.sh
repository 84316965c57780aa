#!/bin/bash
# round 3, batch 82: obs_max_cylinder > 4 (first-design kernels with the 16-wide selection network) and predictor frames of 4-5 chunks (ws kernel): parity;
# profiles of the reset / generator kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab82; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -k "K5 or K8 or K16 or K11 or K7 or error_paths or E300A3C8 or E33A7" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_hip_tp.py -x -q -s -k "matches_oracle or bind_errors" 2>&1 | grep -v "^$" | tail -25
timeout 900 bash tools/profile_envgen.sh r03_envgen > $O/prof_envgen.txt 2>&1; tail -40 $O/prof_envgen.txt
