#!/usr/bin/env python3
"""Soak run: thousands of steps with resets at 65 536 envs in each mode; every buffer must stay finite and sane."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
E, L, STEPS = 65536, 400, int(sys.argv[1]) if len(sys.argv) > 1 else 4000
modes = {"3v1": ({}, {}), "3v1 + predictor": ({}, {"use_TP_net": 1}),
         "3v1 + predictor + obstacles in the frame": ({"use_obstacles": 1, "cylinder": {"max_num": 5, "min_num": 4}}, {"use_TP_net": 1}), "6v2 (extension)": ({"num_agents": 6, "num_targets": 2, "cylinder": {"max_num": 16, "min_num": 8}}, {}),
         # the predictor's one-tile workgroups (small batches) and the wide-frame kernels
         "3v1 + predictor, 2 048 envs": ({"env": {"num_envs": 2048, "max_episode_length": L}}, {"use_TP_net": 1}),
         "6v1 + predictor, 16 384 envs": ({"num_agents": 6, "env": {"num_envs": 16384, "max_episode_length": L}}, {"use_TP_net": 1}),
         "6v2 + predictor + obstacles in the frame": ({"num_agents": 6, "num_targets": 2, "use_obstacles": 1, "cylinder": {"max_num": 16, "min_num": 8}}, {"use_TP_net": 1})}
for name, (task, algo) in modes.items():
    t = {"cylinder": {"max_num": 8, "min_num": 4}, "env": {"num_envs": 65536, "max_episode_length": L}}
    t.update(task)
    E = t["env"]["num_envs"]
    env = HideAndSeek(config.make_cfg(t, algo=algo))
    env.set_seed(1)
    env.reset()
    A = env.num_agents
    acts = [torch.randn(E, A, 4, device=env.device) * s for s in (0.3, 1.0, 3.0, 30.0)]
    t0 = time.perf_counter()
    for i in range(STEPS + 137):            # ends mid-episode
        td = env.step(env.rand_step_input(acts[i % 4]))
        if (i + 1) % L == 0:
            rtd = env.rand_step_input()
            rtd.set("_reset", td[("next", "done")].squeeze(-1))
            env.reset(rtd)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    b = env._bufs
    ok = env.check_finite()
    carried = b["pid_last_rate"][..., 3].clone()          # the carried line-of-sight flag against a fresh evaluation of the final state
    assert env._lib.hns_refresh_derived_state(env._env, env._stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(carried, b["pid_last_rate"][..., 3]), "carried line-of-sight flag differs from a fresh evaluation"
    pos = b["drone_state"][..., :3]
    sp = b["drone_state"][..., 7:10].norm(dim=-1)
    qn = b["drone_state"][..., 3:7].norm(dim=-1)
    tp_ok = True
    if env.use_TP_net:
        tp_ok = all(bool(torch.isfinite(v).all()) for k, v in env._tp_bufs.items() if v.dtype.is_floating_point)
    print(f"{name:18s} {STEPS} steps in {dt:.2f} s: finite {ok and tp_ok}, |pos| max {float(pos.abs().max()):.2f}, speed max {float(sp.max()):.4f}, "
          f"|q| in [{float(qn.min()):.6f}, {float(qn.max()):.6f}], success rate {float(env.stats['success'].mean()):.3f}")
    assert ok and tp_ok and float(sp.max()) <= 1.0 + 1e-5 and abs(float(qn.min()) - 1) < 1e-4 and abs(float(qn.max()) - 1) < 1e-4
