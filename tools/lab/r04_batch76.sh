#!/bin/bash
# round 4, batch 76: the driver's 20-step command with the config-2 leg (14 ms of light device work) run BEFORE the headline region or in its usual place, four runs each, alternating
# (BENCH_CFG2_FIRST was a temporary switch in bench.py: no difference, 17.45-17.61 us either way; removed)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b76; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-traffic-live --tp-steps 0 --abi-steps 0 --no-cpu-baseline --envgen-episodes 0"
for rep in 1 2 3 4; do for first in "" 1; do echo -n "cfg2_first=${first:-0}: "; BENCH_CFG2_FIRST=$first timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'], d['configs']['cfg2']['ms_per_step'], d['configs']['cfg5_shard']['ms_per_step'])"; done; done | tee $O/ab.txt
