#!/bin/bash
# round 5, batch 13: the seeded sweeps at ten times their size on the final library (two-evader prologue, 16-candidate trim), tile mapping forced as well
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_b13; mkdir -p $O
export HNS_FUZZ_SEEDS=2400 HNS_FUZZ_TP_SEEDS=300 HNS_FUZZ_GEN_SEEDS=600
timeout 1500 python -m pytest tests/test_hip_fuzz.py -q -x -k "bit_exact" 2>&1 | tail -3 | tee $O/fuzz_step.txt
HNS_STEP_MAPPING=tile timeout 1500 python -m pytest tests/test_hip_fuzz.py -q -x -k "bit_exact" 2>&1 | tail -3 | tee $O/fuzz_step_tile.txt
timeout 1500 python -m pytest tests/test_hip_tp.py -q -x -k random_configuration 2>&1 | tail -3 | tee $O/fuzz_tp.txt
timeout 1500 python -m pytest tests/test_hip_envgen.py -q -x -k random 2>&1 | tail -3 | tee $O/fuzz_gen.txt
timeout 600 python tools/soak.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/soak.txt
