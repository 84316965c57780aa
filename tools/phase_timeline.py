#!/usr/bin/env python3
"""Absolute timeline of the step kernel's phases by dispatch round (blockIdx / 256): when do the workgroups that
share a CU load, compute and store?  Uses the per-wave phase stamps (hns_set_phase_profile): slot 14/15 are the
chip-wide 100 MHz clock at start/end, the others the shader clock; each wave's stamps are placed on the chip-wide
axis through its own start.  HNS_LIBRARY selects the build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek

E, A, Cn = int(os.environ.get("HNS_TL_ENVS", "65536")), 3, 8
cfg = config.make_cfg({"num_agents": A, "cylinder": {"max_num": Cn, "min_num": Cn}, "env": {"num_envs": E}})
env = HideAndSeek(cfg)
env.reset()
nw = (E // 64) * (A + 1)
buf = torch.zeros(nw, 16, dtype=torch.int64, device=env.device)
act = torch.randn(E, A, 4, device=env.device)
for _ in range(20):
    env.step(env.rand_step_input(act))
env._lib.hns_set_phase_profile(env._env, C.c_void_p(buf.data_ptr()))
env.step(env.rand_step_input(act))
torch.cuda.synchronize()
env._lib.hns_set_phase_profile(env._env, None)
t = buf.cpu().numpy().astype(np.float64).reshape(E // 64, A + 1, 16)
rt0, rt1 = t[..., 14] * 10.0, t[..., 15] * 10.0
z = rt0.min()
dur_c = t[..., 7] - t[..., 0]
f = np.median(dur_c / np.maximum(rt1 - rt0, 1.0))          # shader cycles per ns
print("shader clock ~ %.2f GHz; kernel span %.0f ns" % (f, rt1.max() - z))
marks = [("start", 0), ("loaded", 1), ("p1 done", 2), ("b1 passed", 12), ("p2 done", 3), ("b2 passed", 8), ("obs done", 9), ("3a done", 4), ("b3 passed", 5), ("3b done", 6), ("end", 7)]
rounds = (np.arange(E // 64) >> 8) & 3
print("envs", E)
for role, sl in (("agent waves", slice(0, A)), ("env wave", slice(A, A + 1))):
    print(role + " — median absolute time (ns) of each mark by dispatch round")
    print("%-12s" % "mark" + "".join("%10s" % ("round %d" % r) for r in range(4)))
    for name, k in marks:
        absn = rt0[:, sl] - z + (t[:, sl, k] - t[:, sl, 0]) / f
        ok = t[:, sl, k] > 0
        print("%-12s" % name + "".join("%10.0f" % np.median(absn[rounds == r][ok[rounds == r]]) if ok[rounds == r].any() else "%10s" % "-" for r in range(4)))
