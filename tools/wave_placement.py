#!/usr/bin/env python3
"""Where does the hardware put the step kernel's waves?  Reads HW_ID / XCC_ID of every wave (lab build: -DHNS_LAB,
per-wave slots 10-12 of the phase-profile buffer) and prints, per (XCD, SE, CU, SIMD), how many pursuer and env waves
sit there, plus the blockIdx -> CU pattern.  HNS_LIBRARY selects the build, HNS_LAB_STAGGER the env-wave rotation
(0 = the last wave of every workgroup, n = wave (blockIdx >> (n-1)) % (A+1))."""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import hns_amd  # noqa: F401
from hns_amd import config
from hns_amd.env import HideAndSeek

E, A, Cn = int(os.environ.get("HNS_TL_ENVS", "65536")), int(os.environ.get("HNS_TL_AGENTS", "3")), 8
cfg = config.make_cfg({"num_agents": A, "cylinder": {"max_num": Cn, "min_num": Cn}, "env": {"num_envs": E}})
env = HideAndSeek(cfg)
env.reset()
nw = (E // 64) * (A + 1)
buf = torch.zeros(nw, 16, dtype=torch.int64, device=env.device)
act = torch.randn(E, A, 4, device=env.device)
for _ in range(10):
    env.step(env.rand_step_input(act))
env._lib.hns_set_phase_profile(env._env, C.c_void_p(buf.data_ptr()))
os.environ["HNS_LAB_FLAGS"] = "4096"          # LAB_HWID: slots 10-12 take the placement instead of phase stamps
env.step(env.rand_step_input(act))
torch.cuda.synchronize()
env._lib.hns_set_phase_profile(env._env, None)
t = buf.cpu().numpy().reshape(E // 64, A + 1, 16)
hw, xcc, is_env = t[..., 10], t[..., 11] & 0xF, t[..., 12]
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
print("raw HW_ID of workgroup 0:", [hex(int(x)) for x in hw[0]], "XCC_ID", [int(x) for x in xcc[0]])
print("SIMD of wave w (histogram over workgroups):")
for w in range(A + 1):
    print("  wave %d:" % w, np.bincount(simd[:, w].astype(int), minlength=4).tolist())
load = collections.defaultdict(lambda: [0, 0])
for wg in range(E // 64):
    for w in range(A + 1):
        key = (int(xcc[wg, w]), int(se[wg, w]), int(sh[wg, w]), int(cu[wg, w]), int(simd[wg, w]))
        load[key][int(is_env[wg, w])] += 1
mix = collections.Counter(tuple(v) for v in load.values())
print("SIMDs in use: %d; (pursuer waves, env waves) per SIMD -> number of SIMDs:" % len(load))
for k, n in sorted(mix.items()):
    print("  ", k, n)
cukey = [(int(xcc[wg, 0]), int(se[wg, 0]), int(sh[wg, 0]), int(cu[wg, 0])) for wg in range(E // 64)]
print("CUs in use:", len(set(cukey)))
by_cu = collections.defaultdict(list)
for wg, k in enumerate(cukey):
    by_cu[k].append(wg)
print("workgroups sharing a CU (first 6 CUs):")
for k in sorted(by_cu)[:6]:
    print("  ", k, by_cu[k])
d = collections.Counter()
for k, v in by_cu.items():
    d[tuple(sorted((x - v[0]) for x in v))] += 1
print("blockIdx offsets within a CU -> count:", d.most_common(6))
