#!/bin/bash
# round 4, batch 22: small batches — where a step's time goes (kernel vs the gap between dependent launches; phase stamps of a workgroup)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b22; mkdir -p $O
timeout 200 python tools/small_batch.py 1024 2048 4096 8192 16384 2>&1 | grep "E=" | tee $O/small.txt
timeout 200 rocprofv3 --kernel-trace --stats -d $O/st -- python tools/small_batch.py 2048 4096 --steps=1000 > $O/st.log 2>&1
db=$(ls $O/st/*/*.db | head -1); python tools/rocpd_summary.py "$db" hns_ | tee $O/stats.csv; rm -rf $O/st
timeout 200 python tools/phase_profile.py --envs=4096 --cylinders=5 2>&1 | grep -v amdgpu | tail -16 | tee $O/phase4096.txt
