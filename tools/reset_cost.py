#!/usr/bin/env python3
"""Device time of a full hns_reset at 65 536 envs (raw C-ABI calls, torch events)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
env = HideAndSeek(config.make_cfg({"cylinder": {"max_num": 8, "min_num": 4}, "env": {"num_envs": 65536}}))
env.reset()
s = env._stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5): env._lib.hns_reset(env._env, None, C.c_uint64(1), s)
e0.record()
for _ in range(50): env._lib.hns_reset(env._env, None, C.c_uint64(1), s)
e1.record(); torch.cuda.synchronize()
print(f"hns_reset (all envs) at 65536 envs: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
