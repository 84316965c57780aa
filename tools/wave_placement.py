#!/usr/bin/env python3
"""Which SIMD does each wave of a step-kernel workgroup run on?  (a -DHNS_PROF_HWID build: tools/build_variant.sh hwid -DHNS_PROF_HWID; HNS_LIBRARY=build/variants/libhns_hwid.so)
    python tools/wave_placement.py [--envs=65536 --agents=3 --targets=1 --cylinders=8]
Prints, per wave index of the workgroup (the last one is the env wave), the share of workgroups whose wave sits on SIMD 0..3, and the number of pursuer / env waves per
(CU, SIMD) at the launch's start."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch, hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
a = dict(x[2:].split("=", 1) for x in sys.argv[1:] if x.startswith("--"))
E, A, NT, CYL = int(a.get("envs", 65536)), int(a.get("agents", 3)), int(a.get("targets", 1)), int(a.get("cylinders", 7))
os.environ["HNS_STEP_MAPPING"] = "tile"
env = HideAndSeek(config.make_cfg({"num_agents": A, "num_targets": NT, "cylinder": {"max_num": CYL, "min_num": CYL}, "env": {"num_envs": E, "max_episode_length": 50000}}))
env.reset()
td = env.rand_step_input()
for _ in range(20): env.step(td)
W = A + 1
buf = torch.zeros((E // 64) * W, 16, dtype=torch.int64, device=env.device)
env._lib.hns_set_phase_profile(env._env, C.c_void_p(buf.data_ptr()))
env.step(td)
torch.cuda.synchronize()
env._lib.hns_set_phase_profile(env._env, None)
raw = buf.cpu().numpy().reshape(E // 64, W, 16)
hw = raw[..., 13]
if not hw.any():
    sys.exit("no HW_ID stamps: load a -DHNS_PROF_HWID build with HNS_LIBRARY")
simd = (hw >> 4) & 3
print(f"{E} envs, {A}v{NT}: {E // 64} workgroups of {W} waves")
for w in range(W):
    share = [float((simd[:, w] == s).mean()) for s in range(4)]
    print(f"wave {w} ({'env' if w == W - 1 else 'pursuer'}): SIMD share " + " ".join(f"{x:.2f}" for x in share))
cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
start = raw[..., 14]
first = start <= np.percentile(start, 99 if (E // 64) * W <= 256 * 16 else 45)      # the waves of the launch's first residency round (every wave when one round holds the launch)
heavy = np.zeros((4,)); light = np.zeros((4,))
for s in range(4):
    heavy[s] = ((simd[:, :A] == s) & first[:, :A]).sum(); light[s] = ((simd[:, A] == s) & first[:, A]).sum()
print("first-round waves by SIMD: pursuer", heavy.astype(int).tolist(), " env", light.astype(int).tolist())
# pursuer waves per (XCD, CU, SIMD) among the workgroups of the first residency round (XCD = workgroup index % 8: round-robin dispatch)
xcd = (np.arange(E // 64) % 8)[:, None].repeat(W, 1)
key = (xcd.astype(np.int64) << 16) | (cu.astype(np.int64) << 2) | simd
cnt = {}
for k in key[:, :A][first[:, :A]].ravel().tolist():
    cnt[k] = cnt.get(k, 0) + 1
vals = np.array(list(cnt.values()))
hist = {int(v): int((vals == v).sum()) for v in np.unique(vals)}
print(f"pursuer waves per (XCD, CU, SIMD) in the first round: {hist}  (mean {vals.mean():.2f}, SIMDs seen {len(vals)})")
