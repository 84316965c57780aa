#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab50; mkdir -p $O
for g in 2 4; do
timeout 200 python bench.py --stream-groups $g --steps 400 --warmup 50 --no-cpu-baseline --tp-steps 0 --config-steps 0 --abi-steps 0 > $O/bench_sg$g.json 2>/dev/null
timeout 20 python -c "
import json; d=json.load(open('$O/bench_sg$g.json')); print($g, d['ms_per_step'], d['roofline']['kernel_us'], d['stream_shards'])"
done
