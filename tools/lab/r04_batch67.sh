#!/bin/bash
# round 4, batch 67: predictor with the priority by recurrence step as shipped: parity (predictor tests, two evaders), phase stamps, the bench's predictor leg
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b67; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_tp.py tests/test_two_evaders.py tests/test_manifest.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
timeout 200 python tools/tp_phases.py 65536 2>&1 | grep -v amdgpu | tail -8 | tee $O/tp_phases.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-traffic-live --config-steps 0 --abi-steps 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['tp_mode']; print('tp_mode', t['ms_per_step'], t['observe_us'], t['step_kernel_us'], t['roofline']['frac'])"
