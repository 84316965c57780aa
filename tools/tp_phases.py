#!/usr/bin/env python3
"""Per-wave phase stamps of the weight-stationary predictor kernel (hns_set_phase_profile; the chip-wide 100 MHz clock): where a launch of
hns_tp_lstm_ws_kernel spends its time — prologue (weights, first frames), the T recurrence steps, the output layer, the observation rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch, hns_amd
from hns_amd import config
from hns_amd.env import HideAndSeek
E = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
env = HideAndSeek(config.make_cfg({"cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E, "max_episode_length": 50000}}, algo={"use_TP_net": 1}))
env.reset()
td = env.rand_step_input()
for _ in range(30): env.step(td)
TILES = int(os.environ.get("HNS_TP_TILES", 4))             # column tiles per workgroup (csrc/hns_tp.hip: ws_envs); with the stamps on, four unless forced
nw = ((E + 32 * TILES - 1) // (32 * TILES)) * 8           # workgroups of 32 TILES envs, 8 waves
buf = torch.zeros(max(nw, (E // 64) * 4) , 16, dtype=torch.int64, device=env.device)
env._lib.hns_set_phase_profile(env._env, C.c_void_p(buf.data_ptr()))
env.step(td)
torch.cuda.synchronize()
env._lib.hns_set_phase_profile(env._env, None)
raw = buf.cpu().numpy().astype(np.float64)[:nw]
t = raw[:, :5].reshape(-1, 8, 5) * 10.0     # ns
z = t[..., 0].min()
names = ["start", "prologue done", "recurrence done", "output layer done", "rows done"]
print("mark                 mean ns from the launch's first stamp (min / max over waves)")
for i, n in enumerate(names):
    v = t[..., i] - z
    print("%-20s %9.0f  (%7.0f / %7.0f)" % (n, v.mean(), v.min(), v.max()))
d = np.diff(t, axis=-1)
print("segment means (ns): prologue %.0f, recurrence %.0f, output layer %.0f, rows %.0f; a workgroup's life %.0f (p10 %.0f, p90 %.0f)" % (
    d[..., 0].mean(), d[..., 1].mean(), d[..., 2].mean(), d[..., 3].mean(),
    (t[..., 4].max(1) - t[..., 0].min(1)).mean(), *np.percentile(t[..., 4].max(1) - t[..., 0].min(1), [10, 90])))
if "--loop" in sys.argv:
    # a build with -DTP_WS_LOOP_PROF (tools/build_tp_variant.sh; HNS_LIBRARY=...): shader-clock cycles per wave, summed over the T timesteps
    lp = raw[:, 5:10]
    if lp.sum() == 0:
        print("no loop stamps: load a -DTP_WS_LOOP_PROF build with HNS_LIBRARY")
    else:
        names = ["wait at barrier A (x_t, h_{t-1} published)", "emit frame t+1 / prefetch a window row", "the tiles: operand reads, 15 MFMAs each, cell update",
                 "wait at barrier B (everybody has read h_{t-1})", "publish h_t + loop overhead"]
        tot = lp.sum(1).mean()
        for i, n in enumerate(names):
            print("%-55s %9.0f cycles per launch and wave (%4.1f %%), min %9.0f max %9.0f" % (n, lp[:, i].mean(), 100 * lp[:, i].mean() / tot, lp[:, i].min(), lp[:, i].max()))
        print("sum %9.0f cycles = the recurrence; with the 100 MHz stamps above: %.2f GHz effective shader clock" % (tot, tot / d[..., 1].mean()))
