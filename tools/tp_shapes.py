import sys, time, torch
sys.path.insert(0, "/root/repo")
import hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
for A, Cn in ((3, 8), (6, 16)):
    env = HideAndSeek(config.make_cfg({"num_agents": A, "cylinder": {"max_num": Cn, "min_num": Cn}, "env": {"num_envs": 65536, "max_episode_length": 50000}}, algo={"use_TP_net": 1}))
    env.reset(); td = env.rand_step_input()
    for _ in range(20): env.step(td)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): env.step(td)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    t1 = time.perf_counter()
    for _ in range(200): env._tp_observe()
    torch.cuda.synchronize(); d2 = (time.perf_counter() - t1) / 200
    print(f"A={A} C={Cn}: step+TP {dt*1e6:.1f} us, TP alone {d2*1e6:.1f} us")
