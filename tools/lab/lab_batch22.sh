#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/lab22; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -25 $O/pytest.txt
timeout 600 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --tp-steps 0 --abi-steps 0 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernel_us']); c=d['configs']; print(c['cfg5_shard']['ms_per_step'], c['cfg5_shard']['roofline']['kernel_us']); print(c['cfg4']['ms_per_step'], c['cfg4']['roofline']['kernel_us'], c['cfg4']['generator_ms_per_episode'], c['cfg4']['value_incl_generator_at_800_step_episodes'])"
