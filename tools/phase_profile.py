#!/usr/bin/env python3
"""Per-phase wave timing of the fused step kernel (s_memtime stamps, hns_set_phase_profile).
Run on the GPU box.  Prints mean cycles between phase marks for agent waves and env waves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek

E, A, Cn, NT = 65536, 3, 8, 1
for a in sys.argv[1:]:                                   # e.g. --agents=6 --cylinders=16 --targets=2 (BASELINE config 5's shard)
    if a.startswith("--agents="): A = int(a.split("=")[1])
    if a.startswith("--cylinders="): Cn = int(a.split("=")[1])
    if a.startswith("--targets="): NT = int(a.split("=")[1])
    if a.startswith("--envs="): E = int(a.split("=")[1])
cfg = config.make_cfg({"num_agents": A, "num_targets": NT, "cylinder": {"max_num": Cn, "min_num": Cn}, "env": {"num_envs": E}})
env = HideAndSeek(cfg, write_critic_state="--critic-state" in sys.argv)
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
t16 = buf.cpu().numpy().astype(np.int64).reshape(E // 64, A + 1, 16)
t = t16[..., :8]
t0 = t[..., 0].min()
names = ["load+barrier", "phase1", "wait b1", "phase2", "pub+phase3a", "wait b4", "phase3b", "store"]
d = np.diff(t, axis=-1)
print("kernel span (cycles): %d" % (t[..., 7].max() - t0))
print("block start spread: min %d median %d max %d" % ((t[..., 0] - t0).min(), np.median(t[..., 0] - t0), (t[..., 0] - t0).max()))
print("block duration: median %d  p90 %d" % (np.median(t[:, :, 7].max(1) - t[:, :, 0].min(1)), np.percentile(t[:, :, 7].max(1) - t[:, :, 0].min(1), 90)))
print("%-14s %10s %10s" % ("segment", "agent", "env"))
seg = ["0-1 load", "1-2 phase1", "2-3 b1+phase2", "3-4 b2,b3+3a", "4-5 b4", "5-6 phase3b", "6-7 b5+store"]
for i, n in enumerate(seg):
    print("%-14s %10.0f %10.0f" % (n, d[:, :A, i].mean(), d[:, A, i].mean()))

ag = t16[:, :A, :]
def seg(a, b): return (ag[..., b] - ag[..., a]).mean()
print("agent detail: pid(1->10) %.0f rotor(10->11) %.0f los+term(11->2) %.0f | downwash(2->12) %.0f integrate(12->13) %.0f statestore(13->3) %.0f"
      % (seg(1, 10), seg(10, 11), seg(11, 2), seg(2, 12), seg(12, 13), seg(13, 3)))
print("agent detail: b2+pub+b3(3->8) %.0f obs(8->9) %.0f reward(9->4) %.0f" % (seg(3, 8), seg(8, 9), seg(9, 4)))

ev = t16[:, A, :]
print("env detail: pre-b1 work (0->2) %.0f  b1 wait (2->12) %.0f  post-b1 work (12->3) %.0f | agents: b1 wait (2->12) %.0f" % (
    (ev[..., 2] - ev[..., 0]).mean(), (ev[..., 12] - ev[..., 2]).mean(), (ev[..., 3] - ev[..., 12]).mean(), seg(2, 12)))
rt0, rt1 = t16[..., 14].astype(np.float64) * 10.0, t16[..., 15].astype(np.float64) * 10.0   # ns
z = rt0.min()
print("global clock (ns): first start 0, last start %.0f, first end %.0f, last end %.0f" % (rt0.max() - z, rt1.min() - z, rt1.max() - z))
print("per-block duration (ns): median %.0f p10 %.0f p90 %.0f" % tuple(np.percentile((rt1.max(1) - rt0.min(1)), [50, 10, 90])))
order = np.argsort(rt0.min(1))
print("start time of blocks by dispatch rank (ns): ", [int(rt0.min(1)[order[i]] - z) for i in (0, 255, 256, 511, 512, 767, 768, 1023) if i < len(order)])
