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
for a in sys.argv[1:]:                                   # --mapping=tile|small: force a mapping of the step kernel (default: the library's choice)
    if a.startswith("--mapping="): os.environ["HNS_STEP_MAPPING"] = a.split("=")[1]
for a in sys.argv[1:]:                                   # e.g. --agents=6 --cylinders=16 --targets=2 (BASELINE config 5's shard)
    if a.startswith("--agents="): A = int(a.split("=")[1])
    if a.startswith("--cylinders="): Cn = int(a.split("=")[1])
    if a.startswith("--targets="): NT = int(a.split("=")[1])
    if a.startswith("--envs="): E = int(a.split("=")[1])
cfg = config.make_cfg({"num_agents": A, "num_targets": NT, "cylinder": {"max_num": Cn, "min_num": Cn}, "env": {"num_envs": E}})
env = HideAndSeek(cfg, write_critic_state="--critic-state" in sys.argv)
env.reset()
MAPPING = env.step_mapping                               # small: 2 A + 1 waves per workgroup (owners, env wave, helpers)
WPB = 2 * A + 1 if MAPPING == "small" else A + 1
nw = (E // 64) * WPB
buf = torch.zeros(nw, 16, dtype=torch.int64, device=env.device)
act = torch.randn(E, A, 4, device=env.device)
for _ in range(20):
    env.step(env.rand_step_input(act))
env._lib.hns_set_phase_profile(env._env, C.c_void_p(buf.data_ptr()))
env.step(env.rand_step_input(act))
torch.cuda.synchronize()
env._lib.hns_set_phase_profile(env._env, None)
t16 = buf.cpu().numpy().astype(np.int64).reshape(E // 64, WPB, 16)
# timeline: mean cycles from the workgroup's first stamp to each mark, per role (marks a role does not set are printed as '-')
roles = [("owner" if MAPPING == "small" else "pursuer", slice(0, A)), ("env", slice(A, A + 1))] + ([("helper", slice(A + 1, 2 * A + 1))] if MAPPING == "small" else [])
marks = [(0, "start"), (1, "loaded"), (2, "at b1"), (12, "past b1"), (3, "at b2"), (8, "past b2"), (10, "sweep in"), (9, "own terms"), (4, "at b3"), (5, "past b3"), (11, "tail: calc"), (6, "done")]
blk0 = t16[..., 0].min(axis=1, keepdims=True)
print("%-10s" % "mark" + "".join("%10s" % r for r, _ in roles))
for m, name in marks:
    row = "%-10s" % name
    for _, sl in roles:
        v = t16[:, sl, m]
        row += "%10s" % ("-" if (v == 0).all() else "%.0f" % (v - blk0)[v != 0].mean())
    print(row)
if "--waves" in sys.argv:                                 # the same per wave of the workgroup (mean over workgroups)
    print("%-10s" % "mark" + "".join("%8d" % w for w in range(WPB)))
    for m, name in marks:
        print("%-10s" % name + "".join("%8s" % ("-" if (t16[:, w, m] == 0).all() else "%.0f" % (t16[:, w, m] - blk0[:, 0]).mean()) for w in range(WPB)))
life = t16[..., 7].max(axis=1) - t16[..., 0].min(axis=1)
print("workgroup life (cycles): median %d  p90 %d" % (np.median(life), np.percentile(life, 90)))
rt0, rt1 = t16[..., 14].astype(np.float64) * 10.0, t16[..., 15].astype(np.float64) * 10.0   # ns, the chip-wide 100 MHz clock
z = rt0.min()
print("global clock (ns): first start 0, last start %.0f, first end %.0f, last end %.0f" % (rt0.max() - z, rt1.min() - z, rt1.max() - z))
print("per-workgroup duration (ns): median %.0f p10 %.0f p90 %.0f" % tuple(np.percentile((rt1.max(1) - rt0.min(1)), [50, 10, 90])))
if "--spread" in sys.argv:                                # where the slow workgroups are: by start time, by XCD (workgroup index % 8), by residency slot
    st, en = rt0.min(1) - z, rt1.max(1) - z
    dur = en - st
    q = np.argsort(st)
    n = len(q) // 4
    print("duration (ns) by start-time quartile: " + "  ".join("%.0f (starts %.0f-%.0f)" % (dur[q[i * n:(i + 1) * n]].mean(), st[q[i * n]], st[q[(i + 1) * n - 1]]) for i in range(4)))
    wg = np.arange(len(st))
    print("by XCD (index %% 8): duration " + " ".join("%.0f" % dur[wg % 8 == x].mean() for x in range(8)) + " | end " + " ".join("%.0f" % en[wg % 8 == x].max() for x in range(8)))
    per = max(1, len(st) // 4)
    print("by quarter of the grid (index // %d): start " % per + " ".join("%.0f" % st[wg // per == x].mean() for x in range(4)) + " | duration " + " ".join("%.0f" % dur[wg // per == x].mean() for x in range(4)))
    print("corr(start, duration) = %.2f; last 5%% to end: durations %.0f, starts %.0f (all: %.0f, %.0f)" % (
        np.corrcoef(st, dur)[0, 1], dur[np.argsort(en)[-len(en) // 20:]].mean(), st[np.argsort(en)[-len(en) // 20:]].mean(), dur.mean(), st.mean()))
    seg = [(0, 1, "loads"), (1, 2, "phase 1"), (2, 12, "b1"), (12, 3, "phase 2"), (3, 8, "b2"), (8, 4, "phase 3a"), (4, 5, "b3"), (5, 6, "tail")]
    slow = np.argsort(dur)[-len(dur) // 10:]; fast = np.argsort(dur)[:len(dur) // 10]
    print("pursuer-wave segments (cycles), slowest 10%% of workgroups vs fastest 10%%: " + "  ".join(
        "%s %.0f/%.0f" % (nm, (t16[slow, :A, b] - t16[slow, :A, a]).mean(), (t16[fast, :A, b] - t16[fast, :A, a]).mean()) for a, b, nm in seg))
