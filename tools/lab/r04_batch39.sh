#!/bin/bash
# round 4, batch 39: profiles of the final step kernels (tile mapping 3v1 / 6v2, small mapping at 4 096 envs), then the driver's command and the default bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
KERNEL=hns_step_v4_kernelILi3ELi1ELb0 BYTES=100466688 timeout 500 bash tools/profile_step.sh r04_v4_step_kernel --no-traffic-live
KERNEL=hns_step_v4_kernelILi6ELi2ELb0 BYTES=208207872 timeout 500 bash tools/profile_step.sh r04_step_kernel_a6t2 --agents 6 --cylinders 16 --targets 2 --no-traffic-live
KERNEL=hns_step_small_kernelILi3 BYTES=6131712 timeout 500 bash tools/profile_step.sh r04_step_kernel_small --envs 4096 --cylinders 5 --no-traffic-live
mkdir -p gpurun_out/r04b39
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04b39/bench_driver.json 2> gpurun_out/r04b39/bench_driver.err; python tools/bench_line.py < gpurun_out/r04b39/bench_driver.json | head -14
timeout 900 python bench.py > gpurun_out/r04b39/bench_default.json 2> gpurun_out/r04b39/bench_default.err; python tools/bench_line.py < gpurun_out/r04b39/bench_default.json | head -14
