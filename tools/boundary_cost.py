#!/usr/bin/env python3
"""What an episode boundary costs the step stream (65 536 envs, 3v1, 8 cylinders, 800-step episodes): device time of 64 steps that straddle the masked reset
against 64 plain steps (events on the step stream, the host far ahead of the device both times), and the host time of the `env.reset(td)` call itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hns_amd  # noqa: F401
from hns_amd import config
from hns_amd.env import HideAndSeek
from hns_amd.tensordict_shim import TensorDict

E, L = 65536, 800
env = HideAndSeek(config.make_cfg({"cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E, "max_episode_length": L}}))
env.set_seed(0)
env.reset()
tds = [env.rand_step_input() for _ in range(4)]
rtd = TensorDict({}, [E])
rtd.set("_reset", env._bufs["done"])
env.reset(rtd)                                           # (first masked reset of the process: code-object loads)
ev = lambda: torch.cuda.Event(enable_timing=True)
plain, straddle, host_us = [], [], []
step = 0
for ep in range(6):
    while (step + 32) % L != 0:                           # run up to 32 steps before the boundary
        env.step(tds[step % 4]); step += 1
    a0, a1, b0, b1 = ev(), ev(), ev(), ev()
    # 64 plain steps, measured earlier in the episode on the next round; here: the 64 steps around the boundary
    b0.record()
    for _ in range(32):
        env.step(tds[step % 4]); step += 1
    h0 = time.perf_counter()
    env.reset(rtd)
    host_us.append((time.perf_counter() - h0) * 1e6)
    for _ in range(32):
        env.step(tds[step % 4]); step += 1
    b1.record()
    for _ in range(100):
        env.step(tds[step % 4]); step += 1
    a0.record()
    for _ in range(64):
        env.step(tds[step % 4]); step += 1
    a1.record()
    torch.cuda.synchronize()
    plain.append(a0.elapsed_time(a1) * 1e3)
    straddle.append(b0.elapsed_time(b1) * 1e3)
for ep in range(6):
    print(f"episode {ep}: 64 plain steps {plain[ep]:7.1f} us, 64 steps around the boundary {straddle[ep]:7.1f} us (+{straddle[ep] - plain[ep]:6.1f}); host time of env.reset(td) {host_us[ep]:6.1f} us")
