#!/bin/bash
# round 5, batch 1: whole GPU suite + smoke on the new caller layer / generator pins; dispatch-timestamp trace of the step kernel (launch overlap);
# default bench line as the round's starting point
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_b1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
B="python bench.py --no-cpu-baseline --tp-steps 0 --stream-groups 0 --config-steps 0 --abi-steps 0 --no-traffic-live"
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- $B --steps 600 --warmup 100 > $O/trace.log 2>&1
db=$(ls $O/trace/*/*.db 2>/dev/null | head -1)
if [ -n "$db" ]; then
  python tools/launch_overlap.py "$db" hns_step_v4_kernelILi3ELi1 256 > $O/launch_overlap.txt; head -8 $O/launch_overlap.txt
  python tools/rocpd_summary.py "$db" hns_ > $O/trace_stats.csv; rm -rf $O/trace
fi
grep '^{' $O/trace.log | python tools/bench_line.py 2>/dev/null | head -20
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python tools/bench_line.py < $O/bench_default.json | head -40
