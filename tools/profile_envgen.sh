#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + FETCH/WRITE PMC passes of tools/envgen_profile_run.py (HideAndSeek_envgen at config 4's shape: resets with task
# vectors, perturbation of history tasks, farthest-point trim), every call at one size.
# usage: tools/profile_envgen.sh <tag>
set -u
TAG=${1:-envgen}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
B="python tools/envgen_profile_run.py"
rocprofv3 --kernel-trace --stats -d $OUT/stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VALU SQ_WAVES -d $OUT/pmc1 -- $B > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/pmc2 -- $B > $OUT/pmc2.log 2>&1
for d in stats pmc1 pmc2; do
  db=$(ls $OUT/$d/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py "$db" hns_ > $OUT/$d.csv && rm -rf $OUT/$d
done
python tools/make_profile_txt.py $OUT "${KERNEL:-hns_fps_xcd_kernel}" 0 "$TAG - tools/profile_envgen.sh $TAG" > $OUT/profile.txt
head -6 $OUT/profile.txt; grep -h "hns_" $OUT/stats.csv | head -8
