#!/bin/bash
# round 4, GPU batch 10: the round's profiles (rocprofv3 stats + separate PMC passes), composed by tools/make_profile_txt.py
cd "$GRAFT_REPO_ROOT"
KERNEL=hns_step_v4_kernelILi3ELi1ELb0 BYTES=100466688 timeout 500 bash tools/profile_step.sh r04_v4_step_kernel
KERNEL=hns_step_v4_kernelILi6ELi2ELb0 BYTES=208207872 timeout 500 bash tools/profile_step.sh r04_step_kernel_a6t2 --agents 6 --cylinders 16 --targets 2
timeout 500 bash tools/profile_tp.sh r04
timeout 500 bash tools/profile_envgen.sh r04_envgen
mkdir -p gpurun_out/r04b10
timeout 400 python bench.py --steps 400 --warmup 50 --traffic-live --no-cpu-baseline --tp-steps 0 --config-steps 0 --abi-steps 0 > gpurun_out/r04b10/bench_traffic_live.json 2> gpurun_out/r04b10/bench_traffic_live.err
python tools/bench_line.py < gpurun_out/r04b10/bench_traffic_live.json | head -3
