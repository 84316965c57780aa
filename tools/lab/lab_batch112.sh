#!/bin/bash
# round 3, batch 112: after priming the masked-reset torch path at the first reset: parity subset, smoke, bench without the bench-side priming effect visible
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_torchrl_branch.py tests/test_hip_hover.py -q -x 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 600 python bench.py --no-cpu-baseline --config-steps 0 --tp-steps 0 --abi-steps 0 2>/dev/null | python tools/bench_line.py | head -1
