#!/usr/bin/env python3
"""Per-wave start / staged / loop-end / end stamps of hns_tp_lstm_kernel (100 MHz chip-wide clock)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch, hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
E, A = 65536, 3
env = HideAndSeek(config.make_cfg({"num_agents": A, "cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E, "max_episode_length": 50000}},
                                  algo={"use_TP_net": 1}))
env.reset()
td = env.rand_step_input()
for _ in range(10): env.step(td)
buf = torch.zeros((E // 64) * (A + 1), 16, dtype=torch.int64, device=env.device)
env._lib.hns_set_phase_profile(env._env, C.c_void_p(buf.data_ptr()))
env._tp_observe(); torch.cuda.synchronize()
env._lib.hns_set_phase_profile(env._env, None)
nw = (E // 128) * 8 if "--ws" in sys.argv else (E // 256) * 8
t = buf.cpu().numpy()[:nw, :5].astype(np.float64) * 10.0      # ns
z = t[:, 0].min()
t -= z
print("waves", nw)
print("observation rows (stamp 4 - stamp 3, the tail that was a second kernel): median %.0f ns, max %.0f ns" % (np.median(t[:, 4] - t[:, 3]), (t[:, 4] - t[:, 3]).max()))
for i, n in enumerate(["start", "staged", "loop end", "predictions out", "rows out"]):
    print("%-9s min %8.0f  p10 %8.0f  median %8.0f  p90 %8.0f  max %8.0f ns" % ((n,) + tuple(np.percentile(t[:, i], [0, 10, 50, 90, 100]))))
d = t[:, 4] - t[:, 0]
print("wave lifetime: min %.0f median %.0f max %.0f ns; staging median %.0f ns" % (d.min(), np.median(d), d.max(), np.median(t[:, 1] - t[:, 0])))
wg = t.reshape(-1, 8, 5)
print("per-workgroup end (max over waves): p10 %.0f median %.0f p90 %.0f max %.0f" % tuple(np.percentile(wg[:, :, 4].max(1), [10, 50, 90, 100])))
order = np.argsort(wg[:, :, 4].max(1))
print("slowest workgroups:", order[-8:], "fastest:", order[:8])
if "--ws" in sys.argv:         # -DWS_PHASES build of the weight-stationary kernel: cycles per phase and timestep (timesteps 1..T-1)
    c = buf.cpu().numpy()[:nw, 5:10].astype(np.float64).reshape(-1, 8, 5) / 9.0
    names = ["barrier waits", "next frame", "matrix products (4 tiles)", "cell updates (4 tiles)", "publish h + loop"]
    print("all waves: " + "; ".join("%s %.0f" % (n, c[..., i].mean()) for i, n in enumerate(names)) + "; timestep %.0f cycles" % c.sum(-1).mean())
    for w in range(8):
        print("  wave %d: " % w + " ".join("%6.0f" % c[:, w, i].mean() for i in range(5)))
    hw = buf.cpu().numpy()[:nw, 10].reshape(-1, 8)[:, 0]
    cu = ((hw >> 32) & 0xf) * 64 + ((hw >> 13) & 7) * 16 + ((hw >> 8) & 0xf)       # XCC | SE | CU
    by = {}
    for b, k in enumerate(cu): by.setdefault(int(k), []).append(b)
    pairs = [v for v in by.values() if len(v) == 2]
    print("distinct CUs %d; examples of co-resident workgroups: %s; difference of the two block ids: %s" % (len(by), pairs[:6], sorted(set(abs(a - b) for a, b in pairs))[:8]))
    lt = t[:, 2] - t[:, 1]
    print("recurrence (stamp 2 - stamp 1): median %.0f ns -> %.2f GHz by the cycle counter" % (np.median(lt), c.sum(-1).mean() * 9 / np.median(lt)))
if "--phases" in sys.argv:     # -DTP_PHASES build: cycles per phase summed over the timesteps with a recurrent product
    c = buf.cpu().numpy()[:nw, 5:9].astype(np.float64).reshape(-1, 8, 4) / 9.0
    names = ["rows + frame split", "gate tiles, units 0..31", "gate tiles, units 32..63 (+ cell update 0..31)", "cell update 32..63"]
    for grp, name in ((slice(0, 4), "waves 0-3 (older)"), (slice(4, 8), "waves 4-7 (younger)")):
        g = c[:, grp]
        print(name + ": " + "; ".join("%s %.0f" % (n, g[..., i].mean()) for i, n in enumerate(names)) + "; timestep %.0f cycles" % g.sum(-1).mean())
if "--stamps" in sys.argv:
    c = buf.cpu().numpy()[:nw, :16].astype(np.float64).reshape(-1, 8, 16)
    for grp, name in ((slice(0, 4), "waves 0-3"), (slice(4, 8), "waves 4-7")):
        g = c[:, grp]
        print("%s t=5: M0 %.0f  E0 %.0f  gap %.0f  M1 %.0f  E1 %.0f cyc; timestep 5 %.0f cyc, timestep 6 %.0f cyc" % (
            name, (g[..., 5] - g[..., 4]).mean(), (g[..., 6] - g[..., 5]).mean(), (g[..., 7] - g[..., 6]).mean(), (g[..., 8] - g[..., 7]).mean(),
            (g[..., 9] - g[..., 8]).mean(), (g[..., 11] - g[..., 10]).mean(), (g[..., 12] - g[..., 11]).mean()))
if "--pp" in sys.argv:        # ping-pong stamps (instrumented build only)
    c = buf.cpu().numpy()[:nw, :11].astype(np.float64).reshape(-1, 8, 11)
    for grp, name in ((slice(0, 4), "group 0"), (slice(4, 8), "group 1")):
        g = c[:, grp]
        print("%s (t=5, tj=0): M block %.0f cyc, wait %.0f, E block %.0f, wait %.0f; full cycle %.0f cyc = %.0f ns -> clock %.2f GHz" % (
            name, (g[..., 5] - g[..., 4]).mean(), (g[..., 6] - g[..., 5]).mean(), (g[..., 7] - g[..., 6]).mean(), (g[..., 8] - g[..., 7]).mean(),
            (g[..., 8] - g[..., 4]).mean(), (g[..., 10] - g[..., 9]).mean() * 10, (g[..., 8] - g[..., 4]).mean() / ((g[..., 10] - g[..., 9]).mean() * 10)))
