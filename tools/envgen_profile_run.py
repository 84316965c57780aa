#!/usr/bin/env python3
"""The workload tools/profile_envgen.sh puts under rocprofv3: HideAndSeek_envgen at BASELINE config 4's shape (3v1, 8 cylinders, 65 536 envs),
7 episodes of 100 steps with the whole-batch reset between them — so every call of a kernel in the summary is at ONE size: hns_reset_kernel over
all envs with task vectors, hns_perturb_kernel, hns_fps_kernel / hns_fps_xcd_kernel (the trim of 5 000 + 65 536 tasks), hns_step_v4_kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import hns_amd  # noqa: E402,F401
from hns_amd import config  # noqa: E402
from hns_amd.envgen import HideAndSeek_envgen  # noqa: E402
from hns_amd.tensordict_shim import TensorDict  # noqa: E402

E, L, EP = 65536, 100, 7
cfg = config.make_cfg({"name": "HideAndSeek_envgen", "num_agents": 3, "cylinder": {"max_num": 8, "min_num": 8, "obs_max_cylinder": 3},
                       "env": {"num_envs": E, "max_episode_length": L}, "use_particle_generator": 1, "ratio_unif": 0.3, "eval_iter": 3,
                       "R_min": 0.0, "R_max": 1.0})
env = HideAndSeek_envgen(cfg, headless=True)
env.set_seed(0)
env.reset()
gen = torch.Generator(device=env.device).manual_seed(11)
tds = [TensorDict({"agents": {"action": torch.randn(E, 3, 4, generator=gen, device=env.device)}}, [E]) for _ in range(8)]
rtd = TensorDict({}, [E])
for ep in range(EP):
    for t in range(L):
        env.step(tds[t % 8])
    rtd.set("_reset", env._bufs["done"])
    env.reset(rtd)
torch.cuda.synchronize()
print("history", len(env.gen_buffer))
