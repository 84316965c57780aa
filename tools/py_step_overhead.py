#!/usr/bin/env python3
"""Host-side cost of HideAndSeek.step() (Python class path) vs the raw C-ABI call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
E = 65536
env = HideAndSeek(config.make_cfg({"cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E, "max_episode_length": 100000}}))
env.reset()
td = env.rand_step_input()
for _ in range(50): env.step(td)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): env.step(td)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"env.step(): {dt / 2000 * 1e6:.1f} us/step  ({E * 3 * 2000 / dt:.3e} agent-steps/s)")
