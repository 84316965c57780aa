#!/usr/bin/env python3
"""Cost of the step with the trajectory predictor in the observation (algo.use_TP_net: 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
E = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
for tp in ((1,) if "--tp-only" in sys.argv else (0, 1)):
    env = HideAndSeek(config.make_cfg({"cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E, "max_episode_length": 50000}},
                                      algo={"use_TP_net": tp}))
    env.reset()
    td = env.rand_step_input()
    for _ in range(20): env.step(td)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 40 if "--short" in sys.argv else 200
    for _ in range(n): env.step(td)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"use_TP_net={tp}: {dt / n * 1e6:.1f} us/step  ({E * 3 * n / dt:.3e} agent-steps/s)")
    if tp and "--tp-only" not in sys.argv:
        x = torch.randn(E, 10, 16, device=env.device)
        with torch.no_grad():
            for _ in range(5): env.TP(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): env.TP(x)
            torch.cuda.synchronize(); print(f"  TP_net forward alone: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us")
