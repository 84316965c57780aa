#!/usr/bin/env python3
"""Cost of the optional ray-fan range sensor (hns_raycast) at 65 536 envs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
E = 65536
env = HideAndSeek(config.make_cfg({"cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E}}))
env.reset()
for rays in (16, 64):
    for _ in range(5): env.raycast(rays, 2.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): env.raycast(rays, 2.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    print(f"hns_raycast {rays} rays x 3 pursuers x {E} envs: {dt*1e6:.1f} us ({E*3*rays/dt:.3e} rays/s)")
