#!/usr/bin/env python3
"""Small batches (BASELINE config 2 and the reference's own 2 048-env default): step time through env.step, and — under
`rocprofv3 --kernel-trace --stats` — the step kernel's own duration, i.e. how much of a step is the gap between two dependent launches.
usage: small_batch.py [envs ...] [--steps=N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
from hns_amd.tensordict_shim import TensorDict

sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [2048, 4096]
steps, A, NT, CYL = 2000, 3, 1, None                     # --agents= --targets= --cylinders=N (N active cylinders; default: config 2's 5 inactive slots)
for a in sys.argv[1:]:
    if a.startswith("--steps="): steps = int(a.split("=")[1])
    if a.startswith("--agents="): A = int(a.split("=")[1])
    if a.startswith("--targets="): NT = int(a.split("=")[1])
    if a.startswith("--cylinders="): CYL = int(a.split("=")[1])
for E in sizes:
    cyl = {"max_num": 5, "min_num": 0, "fixed_num": 0} if CYL is None else {"max_num": CYL, "min_num": CYL}
    cfg = config.make_cfg({"num_agents": A, "num_targets": NT, "cylinder": cyl, "env": {"num_envs": E}})
    env = HideAndSeek(cfg)
    env.reset()
    tds = [TensorDict({"agents": {"action": torch.randn(E, A, 4, device=env.device)}}, [E]) for _ in range(8)]
    for i in range(100):
        env.step(tds[i % 8])
    torch.cuda.synchronize()
    env.region_begin()
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(tds[i % 8])
    env.region_end()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"E={E}: {1e6 * (t1 - t0) / steps:.2f} us per step (wall), {1e3 * env.region_ms() / steps:.2f} us per step (events around the region)")
