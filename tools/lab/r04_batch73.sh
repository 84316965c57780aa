#!/bin/bash
# round 4, batch 73: what the episode boundaries cost the headline region (4 000 steps with 800-step episodes against one endless episode)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b73; mkdir -p $O
B="python bench.py --steps 4000 --warmup 50 --no-traffic-live --config-steps 0 --tp-steps 0 --abi-steps 0 --no-cpu-baseline"
for rep in 1 2; do for ep in 800 1000000 200; do echo -n "episode $ep: "; timeout 300 $B --episode $ep 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'])"; done; done | tee $O/ep.txt
