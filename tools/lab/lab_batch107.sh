#!/bin/bash
# round 3, batch 107: merge shortcut test + host cost again
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_torchrl_branch.py tests/test_manifest.py -q -x -k "merge or torchrl or manifest or graph" 2>&1 | tail -4
timeout 300 python - <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
import hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
from hns_amd.tensordict_shim import TensorDict
E = 4096
env = HideAndSeek(config.make_cfg({"num_agents": 3, "cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E, "max_episode_length": 100000}}))
env.reset()
tds = [TensorDict({"agents": {"action": torch.randn(E, 3, 4, device=env.device)}}, [E]) for _ in range(8)]
for i in range(200): env.step(tds[i % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5000): env.step(tds[i % 8])
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"E={E}: host loop {(t1 - t0) / 5000 * 1e6:.2f} us per env.step call")
PY
