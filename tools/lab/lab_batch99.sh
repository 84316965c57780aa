#!/bin/bash
# round 3, batch 99: the seeded sweeps at ten times their size (one-off deep run)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HNS_FUZZ_SEEDS=3000 HNS_FUZZ_TP_SEEDS=600 HNS_FUZZ_GEN_SEEDS=300
timeout 1500 python -m pytest tests/test_hip_fuzz.py -q -x -k "bit_exact" 2>&1 | tail -6
timeout 1500 python -m pytest tests/test_hip_tp.py -q -x -k random_configuration 2>&1 | tail -6
timeout 1500 python -m pytest tests/test_hip_envgen.py -q -x -k random 2>&1 | tail -6
