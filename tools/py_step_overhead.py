#!/usr/bin/env python3
"""Host cost of one `env.step(td)` call of the Python class (no GPU wait: the loop is timed up to the last launch, then
synchronised) beside the kernel time: the class path must stay below the kernel's ~25 us per launch to be free."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
from hns_amd.tensordict_shim import TensorDict

for E in (1024, 65536):
    cfg = config.make_cfg({"num_agents": 3, "cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E}})
    env = HideAndSeek(cfg)
    env.reset()
    tds = [TensorDict({"agents": {"action": torch.randn(E, 3, 4, device=env.device)}}, [E]) for _ in range(8)]
    for i in range(200):
        env.step(tds[i % 8])
    torch.cuda.synchronize()
    n = 3000
    t0 = time.perf_counter()
    for i in range(n):
        env.step(tds[i % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"E={E}: host {1e6 * (t1 - t0) / n:.2f} us per env.step call issued, {1e6 * (t2 - t0) / n:.2f} us per step completed")
