#!/usr/bin/env python3
"""Which hidden units deviate?  The output layer is set to pick single hidden units (row r <- unit u0 + step r), fill mode, one call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import ctypes as C
import numpy as np, torch, hns_amd
from hns_amd import abi, config
from hns_amd.env import HideAndSeek
import hns_oracle as O
E = 512
np.set_printoptions(precision=2, linewidth=250)
for u0, step in ((0, 4), (1, 4), (2, 1), (40, 1)):
    cfg = config.make_cfg({"num_agents": 3, "cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E, "max_episode_length": 800}}, algo={"use_TP_net": 1})
    env = HideAndSeek(cfg, headless=True)
    units = [u0 + step * r for r in range(15)]
    with torch.no_grad():
        sd = env.TP.state_dict()
        sd["fc.weight"].zero_(); sd["fc.bias"].zero_()
        for r, u in enumerate(units):
            sd["fc.weight"][r, u] = 1.0
    env.set_seed(0); env.reset()
    g = torch.Generator(device="cpu").manual_seed(3)
    for t in range(6):
        env.step(env.rand_step_input(torch.randn(E, 3, 4, generator=g).to(env.device)))
    tpa = {k: v.cpu().numpy().copy() for k, v in env._tp_bufs.items() if k != "packed"}
    tpa["packed"] = np.zeros(16, np.uint8)
    for f, key in abi.TP_STATE_DICT_KEYS.items():
        tpa[f] = env.TP.state_dict()[key].detach().cpu().numpy().copy()
    win = env._tp_bufs["history"].clone()
    env._tp_bufs["history"].copy_(win)
    assert env._lib.hns_tp_observe(env._env, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    hip = env._tp_bufs["pred"].cpu().numpy().reshape(E, 15).copy()
    tpa["history"] = win.cpu().numpy().copy()
    O.tp_observe(env.hcfg, env.export_state(), tpa, fill=False)
    d = np.abs(hip - tpa["pred"].reshape(E, 15))
    bad = np.nonzero(d.max(1) > 1e-6)[0]
    print(f"units {units}: max {d.max():.1e}; bad envs {bad[:10]}")
    for b in bad[:5]:
        print(f"   env {b}: per-unit error {d[b]}")
