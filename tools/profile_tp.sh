#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + one PMC pass of the TP-mode step.
# usage: tools/profile_tp.sh <tag>
set -u
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tp_$TAG
mkdir -p $OUT
# the stats pass profiles the bench's own predictor leg (2 000 steps + the hns_tp_observe repetitions its `observe_us` is taken from), so the line and
# this file describe the same launches; the counter pass a short run of the same leg
B="python bench.py --no-cpu-baseline --no-traffic-live --config-steps 0 --abi-steps 0 --stream-groups 0 --steps 100 --warmup 20"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -- $B --tp-steps 2000 > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc1 -- $B --tp-steps 60 > $OUT/pmc1.log 2>&1
for d in stats pmc1; do
  db=$(ls $OUT/$d/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py "$db" hns_ > $OUT/$d.csv && rm -rf $OUT/$d
done
grep -h '^{' $OUT/stats.log | python tools/bench_line.py | grep tp_mode
# useful FLOP per launch of hns_tp_observe at 65 536 envs, 16-value frames, T = 10, F = 5: 2 (T 4 64 (I + 64) + 64 3F) x 65 536
python tools/make_profile_txt.py $OUT "${KERNEL:-hns_tp_lstm_ws_kernelILi1ELi4}" "flop:${FLOP:-2.697e10}" "$TAG - tools/profile_tp.sh $TAG" > $OUT/profile.txt
head -8 $OUT/profile.txt
