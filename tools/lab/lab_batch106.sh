#!/bin/bash
# round 3, batch 106: per-step host cost of env.step after the merge shortcut (4 096-env steps: the kernel is 10 us, the loop shows the host), GPU tests of the caller-facing paths, bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/lab106; mkdir -p $O
timeout 900 python -m pytest tests/test_torchrl_branch.py tests/test_manifest.py tests/test_hip_parity.py tests/test_bench_contract.py -q -x 2>&1 | tail -4
timeout 300 python - <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
import hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
from hns_amd.tensordict_shim import TensorDict
for E in (64, 4096):
    env = HideAndSeek(config.make_cfg({"num_agents": 3, "cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E, "max_episode_length": 100000}}))
    env.reset()
    tds = [TensorDict({"agents": {"action": torch.randn(E, 3, 4, device=env.device)}}, [E]) for _ in range(8)]
    for i in range(200): env.step(tds[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5000): env.step(tds[i % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"E={E}: host loop {(t1 - t0) / 5000 * 1e6:.2f} us per env.step call (queue drained {(t2 - t1) * 1e3:.2f} ms later)")
PY
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python tools/bench_line.py < $O/bench.json | head -3
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python tools/bench_line.py < $O/bench_driver.json | head -1
