#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_bench; mkdir -p $O
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/time.txt; tail -3 $O/time.txt
HNS_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 400 --warmup 50 > $O/bench_2ranks_gloo.json 2> $O/bench_2.err
python bench.py --stream-groups 2 --steps 400 --warmup 50 --no-cpu-baseline --tp-steps 0 --config-steps 0 > $O/bench_streams.json 2>/dev/null
python -c "
import json
for f in ('bench_default','bench_2ranks_gloo','bench_streams'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['n_gpus'], d['roofline']['kernel_us'], d['roofline']['frac'], d.get('stream_shards') and d['stream_shards']['ms_per_step'])
"
