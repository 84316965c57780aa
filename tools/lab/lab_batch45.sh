#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab45; mkdir -p $O
timeout 300 python bench.py --no-cpu-baseline --stream-groups 0 --config-steps 0 --abi-steps 0 --steps 300 --warmup 50 --tp-steps 300 > $O/bench.json 2> $O/bench.err
timeout 20 python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['frac']); print(d['tp_mode'])"
