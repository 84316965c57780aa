#!/bin/bash
# round 4, GPU batch 4: several samples per exchange in the farthest-point trim (bit-identical indices; A/B of the time), the bench loop
# with two HIP shards against the whole batch, step + predictor as staggered shards
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b4
( timeout 900 python -m pytest tests/test_hip_envgen.py tests/test_envgen.py tests/test_bench_contract.py -m gpu -q -p no:cacheprovider ) > gpurun_out/r04b4/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r04b4/pytest.log | cut -c1-220 | head -30
for b in 1 2 3 4; do echo "HNS_FPS_BATCH=$b"; HNS_FPS_BATCH=$b timeout 200 python tools/fps_time.py 2>&1 | tail -6; done | tee gpurun_out/r04b4/fps_time.txt
timeout 400 python tools/tp_overlap_lab.py 65536 300 2>&1 | tee gpurun_out/r04b4/tp_overlap.txt | tail -8
timeout 400 python bench.py > gpurun_out/r04b4/bench_default.json 2> gpurun_out/r04b4/bench_default.err
python tools/bench_line.py < gpurun_out/r04b4/bench_default.json 2>&1 | head -20
