#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab52; mkdir -p $O
timeout 300 python bench.py --no-cpu-baseline --stream-groups 0 --tp-steps 0 --abi-steps 0 --steps 300 --warmup 50 > $O/bench.json 2> $O/bench.err
timeout 20 python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['roofline']['kernel_us']); c=d['configs']['cfg4']; print({k:c[k] for k in c if k!='roofline' and k!='workload'})"
HNS_FPS_KERNEL=chip timeout 300 python bench.py --no-cpu-baseline --stream-groups 0 --tp-steps 0 --abi-steps 0 --steps 300 --warmup 50 > $O/bench_chip.json 2> $O/bench_chip.err
timeout 20 python -c "
import json; d=json.load(open('$O/bench_chip.json')); c=d['configs']['cfg4']; print('chip', {k:c[k] for k in c if k!='roofline' and k!='workload'})"
timeout 300 python -m pytest tests/test_envgen.py tests/test_hip_envgen.py -m gpu -x -q 2>&1 | tail -3
