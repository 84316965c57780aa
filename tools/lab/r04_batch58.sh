#!/bin/bash
# round 4, batch 58: the predictor leg of bench.py at 200 / 1000 / 3000 timed steps (is 110 us against 100 us in longer runs a matter of region length?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b58; mkdir -p $O
for n in 200 1000 3000 200; do echo -n "tp-steps $n: "; timeout 600 python bench.py --steps 20 --warmup 5 --no-traffic-live --config-steps 0 --abi-steps 0 --no-cpu-baseline --tp-steps $n 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['tp_mode']; print(t['ms_per_step'], t['observe_us'], t['step_kernel_us'])"; done | tee $O/tp.txt
