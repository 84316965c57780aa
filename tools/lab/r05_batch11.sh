#!/bin/bash
# round 5, batch 11: the two-evader step kernel with all first loads issued before the first wait (global, not flat, cylinder loads; the integrator's quad last)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_b11; mkdir -p $O
timeout 900 python -m pytest tests/test_two_evaders.py tests/test_hip_parity.py tests/test_hip_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 120 python tools/phase_profile.py --agents=6 --cylinders=16 --targets=2 2>&1 | grep -v amdgpu.ids | tee $O/phase_a6t2.txt
timeout 300 python tools/ab_env.py HNS_LIBRARY=build/variants/libhns_prologue_before.so HNS_LIBRARY=multi-uav-pursuit-evasion_amd/libhns.so 65536 --agents=6 --targets=2 --cylinders=16 --steps=1000 --blocks=4 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
