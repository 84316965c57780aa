#!/bin/bash
# round 4, GPU batch 3: the predictor's new cell update (A/B against the round-3 form), the fma_mix split check, the reset_pid golden on the
# GPU, the suites that failed in batch 2, step + predictor as shards on two streams
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b3
timeout 60 ./build/microbench/fma_mix_check 2>&1 | tee gpurun_out/r04b3/fma_mix_check.txt
( timeout 900 python -m pytest tests/test_reset_pid.py tests/test_hip_tp.py tests/test_envgen.py tests/test_two_evaders.py tests/test_tp_net.py -m gpu -q -p no:cacheprovider ) > gpurun_out/r04b3/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r04b3/pytest.log | cut -c1-220 | head -30
timeout 400 python tools/tp_lab.py --rounds=5 new=multi-uav-pursuit-evasion_amd/libhns.so v1=build/variants/libhns_cellv1.so new2=multi-uav-pursuit-evasion_amd/libhns.so 2>&1 | tee gpurun_out/r04b3/tp_lab.txt | tail -12
timeout 400 python tools/tp_overlap_lab.py 65536 300 2>&1 | tee gpurun_out/r04b3/tp_overlap.txt | tail -8
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04b3/bench_driver.json 2> gpurun_out/r04b3/bench_driver.err
python tools/bench_line.py < gpurun_out/r04b3/bench_driver.json 2>&1 | head -20
