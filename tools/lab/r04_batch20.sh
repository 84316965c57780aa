#!/bin/bash
# round 4, GPU batch 20: k + 1 tracked keys in the fixed-shape instantiations — parity suites, then the round's step-kernel profiles on the final kernels
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b20
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_two_evaders.py tests/test_reset_pid.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4
KERNEL=hns_step_v4_kernelILi3ELi1ELb0 BYTES=100466688 timeout 500 bash tools/profile_step.sh r04_v4_step_kernel --no-traffic-live
KERNEL=hns_step_v4_kernelILi6ELi2ELb0 BYTES=208207872 timeout 500 bash tools/profile_step.sh r04_step_kernel_a6t2 --agents 6 --cylinders 16 --targets 2 --no-traffic-live
timeout 300 python bench.py --no-cpu-baseline --tp-steps 0 --abi-steps 0 --no-traffic-live > gpurun_out/r04b20/bench.json 2>/dev/null; python tools/bench_line.py < gpurun_out/r04b20/bench.json | head -8
