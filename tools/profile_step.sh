#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# usage: tools/profile_step.sh <tag> [extra bench args]
set -u
TAG=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-traffic-live --tp-steps 0 --stream-groups 0 --config-steps 0 --abi-steps 0 $*"
# the stats pass: the bench's default region (2 000 steps after 200 of warm-up), nothing riding on the launches
rocprofv3 --kernel-trace --stats -d $OUT/stats -- $B --steps 2000 --warmup 200 > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc1 -- $B --steps 60 --warmup 10 > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc2 -- $B --steps 60 --warmup 10 > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $OUT/pmc3 -- $B --steps 60 --warmup 10 > $OUT/pmc3.log 2>&1
db=$(ls $OUT/stats/*/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/launch_overlap.py "$db" "${KERNEL:-hns_step_v4_kernelILi3ELi1}" 256 > $OUT/launch_overlap.txt
for d in stats pmc1 pmc2 pmc3; do
  db=$(ls $OUT/$d/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py "$db" hns_ > $OUT/$d.csv && rm -rf $OUT/$d
done
# the profile file: tables + a header computed from them (KERNEL / BYTES name the kernel and its algorithmic bytes per launch)
python tools/make_profile_txt.py $OUT "${KERNEL:-hns_step_v4_kernelILi3ELi1}" "${BYTES:-100466688}" "$TAG - tools/profile_step.sh $TAG $*" > $OUT/profile.txt
head -12 $OUT/profile.txt
