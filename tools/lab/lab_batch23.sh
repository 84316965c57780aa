#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab23; mkdir -p $O
B=build/lab
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 1200 python tools/step_lab.py --rounds=3 v3g=$B/libhns_v3g2.so v3h=$B/libhns_v3h.so v3g_b=$B/libhns_v3g2.so v3h_b=$B/libhns_v3h.so > $O/step_lab.txt 2>&1
cat $O/step_lab.txt
BB="python bench.py --no-cpu-baseline --tp-steps 0 --stream-groups 0 --config-steps 0 --abi-steps 0 --steps 60 --warmup 10"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc2 -- $BB > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc3 -- $BB > $O/pmc3.log 2>&1
for d in pmc2 pmc3; do db=$(ls $O/$d/*/*.db 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_summary.py "$db" hns_step > $O/$d.csv && rm -rf $O/$d; done
grep -h "SIZE" $O/pmc2.csv $O/pmc3.csv
