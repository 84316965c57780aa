#!/bin/bash
# round 4, GPU batch 18: experiment — cylinder count and k as compile-time constants in the step kernel (A/B/A/B on one box)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b18
B="python bench.py --steps 3000 --warmup 200 --no-cpu-baseline --tp-steps 0 --config-steps 0 --abi-steps 0 --no-traffic-live"
for rep in 1 2; do
  for v in product fixc8; do
    L=multi-uav-pursuit-evasion_amd/libhns.so; [ $v = fixc8 ] && L=build/variants/libhns_fixc8.so
    HNS_LIBRARY=$L timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); r=d['roofline']; print('3v1 C8  $v', d['ms_per_step'], r['kernel_us_dispatch_events'], r['kernel_us_post_region'])"
  done
  for v in product fixc16; do
    L=multi-uav-pursuit-evasion_amd/libhns.so; [ $v = fixc16 ] && L=build/variants/libhns_fixc16.so
    HNS_LIBRARY=$L timeout 200 $B --agents 6 --cylinders 16 --targets 2 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); r=d['roofline']; print('6v2 C16 $v', d['ms_per_step'], r['kernel_us_dispatch_events'], r['kernel_us_post_region'])"
  done
done 2>&1 | tee gpurun_out/r04b18/fixc.txt
