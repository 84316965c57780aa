#!/usr/bin/env python3
"""A/B of two library settings inside ONE process, alternating timed blocks (the pool's boxes drift by a few percent from process to process
and with the clock state, which is the size of the effects under test):
  ab_env.py VAR=a VAR=b [envs] [--agents= --targets= --cylinders= --steps= --blocks=]      e.g.  ab_env.py HNS_STEP_PRIO=0 HNS_STEP_PRIO=1 65536   (--tp: with the predictor)
Each setting gets its own env (the variable is read by hns_create); prints the median / min of the per-block step times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import hns_amd
from hns_amd import abi, config
from hns_amd.env import HideAndSeek
from hns_amd.tensordict_shim import TensorDict

settings = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [65536]
steps, blocks, A, NT, CYL = 3000, 7, 3, 1, 8
TP = "--tp" in sys.argv                                   # with the trajectory predictor in the observation (algo.use_TP_net: 1)
for a in sys.argv[1:]:
    if a.startswith("--steps="): steps = int(a.split("=")[1])
    if a.startswith("--blocks="): blocks = int(a.split("=")[1])
    if a.startswith("--agents="): A = int(a.split("=")[1])
    if a.startswith("--targets="): NT = int(a.split("=")[1])
    if a.startswith("--cylinders="): CYL = int(a.split("=")[1])
for E in sizes:
    envs = []
    for sset in settings:
        k, v = sset.split("=", 1)
        os.environ[k] = v
        abi._LIB = None                                      # (HNS_LIBRARY=<another build>: every env loads the library its setting names)
        task = {"num_agents": A, "num_targets": NT, "cylinder": {"max_num": CYL, "min_num": CYL},
                "env": {"num_envs": E, "max_episode_length": 50000 if TP else 1000000}}
        if k.startswith("task."):                            # a task option instead of an environment variable: task.tp_overlap=1
            task[k[5:]] = int(v)
        cfg = config.make_cfg(task, algo={"use_TP_net": 1 if TP else 0})
        e = HideAndSeek(cfg)
        e.reset()
        envs.append(e)
        del os.environ[k]
    tds = [TensorDict({"agents": {"action": torch.randn(E, A, 4, device=envs[0].device)}}, [E]) for _ in range(8)]
    t = [[] for _ in envs]
    for b in range(blocks + 1):
        for i, e in enumerate(envs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(steps):
                e.step(tds[s % 8])
            torch.cuda.synchronize()
            if b:                                            # block 0 warms up
                t[i].append((time.perf_counter() - t0) / steps * 1e6)
    for sset, ti in zip(settings, t):
        print(f"E={E} A={A} NT={NT} C={CYL}  {sset:24s} median {np.median(ti):7.2f} us per step  (min {min(ti):.2f}, max {max(ti):.2f}; {blocks} blocks of {steps})")
