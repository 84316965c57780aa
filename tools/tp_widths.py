#!/usr/bin/env python3
"""hns_tp_observe alone (events around 200 launches) at the predictor's frame widths: 1 chunk (3 pursuers), 2 chunks (3 pursuers + 5 cylinders in the
frame, the reference's use_obstacles shape: 31 values), 3 chunks (3 + 8: 40 values), 4 (6 + 12: 61) and 5 (6 + 16: 73).  HNS_TP_KERNEL=tile|ws forces
a kernel where both exist (read once per process: run this once per setting).  usage: tp_widths.py [envs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hns_amd  # noqa: F401,E402
from hns_amd import config  # noqa: E402
from hns_amd.env import HideAndSeek  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for A, Cn, obst in ((3, 8, 0), (3, 5, 1), (3, 8, 1), (6, 12, 1), (6, 16, 1)):
    env = HideAndSeek(config.make_cfg({"num_agents": A, "use_obstacles": obst, "cylinder": {"max_num": Cn, "min_num": Cn}, "env": {"num_envs": E, "max_episode_length": 50000}},
                                      algo={"use_TP_net": 1}))
    env.reset()
    td = env.rand_step_input()
    for _ in range(12):
        env.step(td)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        env._tp_observe()
    e0.record()
    for _ in range(200):
        env._tp_observe()
    e1.record()
    torch.cuda.synchronize()
    print(f"HNS_TP_KERNEL={os.environ.get('HNS_TP_KERNEL', '(default)')}: A={A} C={Cn} use_obstacles={obst} frame {env.tp_frame_dim} values "
          f"({(env.tp_frame_dim + 15) // 16} chunks): hns_tp_observe {e0.elapsed_time(e1) / 200 * 1e3:.1f} us", flush=True)
    del env
    torch.cuda.empty_cache()
