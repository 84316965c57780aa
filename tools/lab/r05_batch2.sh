#!/bin/bash
# round 5, batch 2: predictor kernels by frame width (default choice, tile forced, ws forced); bench contract tests after the roofline rework
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_b2; mkdir -p $O
for k in "" tile ws; do
  HNS_TP_KERNEL=$k timeout 200 python tools/tp_widths.py 65536 >> $O/tp_widths.txt 2>&1
done
cat $O/tp_widths.txt | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_bench_contract.py tests/test_envgen.py tests/test_hip_envgen.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
