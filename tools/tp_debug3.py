#!/usr/bin/env python3
"""-DWS_DEBUG build: pre-activations / cell state / h of one (env, unit) per timestep against an fp64 evaluation of the same window."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import ctypes as C
import numpy as np, torch, hns_amd
from hns_amd import abi, config
from hns_amd.env import HideAndSeek
E, ENV, UNIT = 512, int(sys.argv[1]), int(sys.argv[2])
np.set_printoptions(precision=8, linewidth=250, suppress=False)
cfg = config.make_cfg({"num_agents": 3, "cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E, "max_episode_length": 800}}, algo={"use_TP_net": 1})
# same parameter / state sequence as tp_debug2's fourth configuration
for u0, step in ((0, 4), (1, 4), (2, 1), (40, 1)):
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
win = env._tp_bufs["history"].clone()
buf = torch.zeros(1 << 16, dtype=torch.int64, device=env.device)
env._lib.hns_set_phase_profile(env._env, C.c_void_p(buf.data_ptr()))
assert env._lib.hns_tp_observe(env._env, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
torch.cuda.synchronize()
env._lib.hns_set_phase_profile(env._env, None)
dbg = buf.view(torch.float32).cpu().numpy()[4096:4096 + 80].reshape(10, 8)
sd = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in env.TP.state_dict().items()}
X = win.cpu().numpy().astype(np.float64)[ENV]        # the call above shifted in place from `win`: window = rows 1.. + the new frame
Xn = env._tp_bufs["history"].cpu().numpy().astype(np.float64)[ENV]
Wih, Whh, b = sd["lstm.weight_ih_l0"], sd["lstm.weight_hh_l0"], sd["lstm.bias_ih_l0"] + sd["lstm.bias_hh_l0"]
h = np.zeros(64); c = np.zeros(64)
sig = lambda z: 1 / (1 + np.exp(-z))
L2E = 1.4426950408889634
print("t | z_i z_f z_g z_o (HIP, unscaled back) vs fp64 | c | h")
for t in range(10):
    z = Wih @ Xn[t] + Whh @ h + b
    i, f, gg, o = sig(z[:64]), sig(z[64:128]), np.tanh(z[128:192]), sig(z[192:])
    c = f * c + i * gg; h = o * np.tanh(c)
    zh = np.array([dbg[t, 0] / -L2E, dbg[t, 1] / -L2E, dbg[t, 2] / (-2 * L2E), dbg[t, 3] / -L2E])
    zr = np.array([z[UNIT], z[64 + UNIT], z[128 + UNIT], z[192 + UNIT]])
    print(t, "z hip", zh, "ref", zr, "dz", zh - zr, "| c", dbg[t, 4], c[UNIT], "| h", dbg[t, 5], h[UNIT], "dh %.2e" % (dbg[t, 5] - h[UNIT]), "| split hi %.9e lo %.9e (h - hi = %.9e)" % (dbg[t, 6], dbg[t, 7], np.float32(dbg[t, 5]) - np.float32(dbg[t, 6])))
