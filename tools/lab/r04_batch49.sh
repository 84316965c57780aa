#!/bin/bash
# round 4, batch 49: the driver's 20-step command: copy yardstick before / after the region x priority boost on / off, three runs each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b49; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-traffic-live --config-steps 0 --tp-steps 0 --abi-steps 0 --no-cpu-baseline"
for rep in 1 2 3; do for ca in "" 1; do for pr in 0 1; do
  echo -n "copy_after=${ca:-0} prio=$pr: "; BENCH_COPY_AFTER=$ca HNS_STEP_PRIO=$pr timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_us_post_region'))"
done; done; done 2>&1 | tee $O/ab.txt
