#!/usr/bin/env python3
"""BASELINE config 4: HideAndSeek_envgen at 65 536 envs — step rate and per-episode generator cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hns_amd
from hns_amd import config
from hns_amd.envgen import HideAndSeek_envgen

E, L, EPISODES = 65536, int(os.environ.get("EP_LEN", "100")), 7
cfg = config.make_cfg({"name": "HideAndSeek_envgen", "num_agents": 3, "ratio_unif": 0.3, "eval_iter": 3, "R_min": 0.0, "R_max": 1.0,
                       "use_particle_generator": 1, "cylinder": {"max_num": 8, "min_num": 8},
                       "env": {"num_envs": E, "max_episode_length": L}})
env = HideAndSeek_envgen(cfg)
env.set_seed(0)
env.reset()
acts = [torch.randn(E, 3, 4, device=env.device) for _ in range(4)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for ep in range(EPISODES):
    g0 = env.generator_seconds
    for t in range(L):
        td = env.step(env.rand_step_input(acts[t % 4]))
    rtd = env.rand_step_input()
    rtd.set("_reset", td[("next", "done")].squeeze(-1))
    env.reset(rtd)
    torch.cuda.synchronize()
    print(f"episode {ep}: generator {1e3 * (env.generator_seconds - g0):8.1f} ms  history {len(env.gen_buffer):5d}  num_unif {env.num_unif}")
dt = time.perf_counter() - t0
print(f"total {dt:.2f} s for {EPISODES} episodes x {L} steps: {E * 3 * L * EPISODES / dt:.3e} agent-steps/s incl. generator")
