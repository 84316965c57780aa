"""Free-running parity (golden S_1 .. S_100).  tests/golden/g_episode_free_*.npz hold 100 consecutive states of the
reference's closed loop (every stage reference code, the integrator = the A5 spec) from a stored S_0 with stored actions.  Here the
oracle (CPU) and the HIP env (`-m gpu`) start at S_0 and step ON THEIR OWN STATE with those actions — no teacher forcing — and the drift
per field is reported per step.

Bounds: pursuer position / attitude / velocity within 1e-5 of the golden at EVERY one of the 100 steps; everything within 1e-5 for the
first 10 steps (evader included); `done` bit-exact at every step.  Looser, stated: the evader's velocity is v_prey * F / (|F| + 1e-5)
PER AXIS (hideandseek.py:741) — where an axis of the potential-field force passes through zero its sign, hence a 2 * v_prey * dt = 2.6 cm
step of the evader, hangs on the last bit of F, so single envs part from the golden trajectory late in the episode: evader position and
the distance reward within 1e-5 for at least 75 % of the envs at step 100 (measured: 13 of 16 and 8 of 8) and within 5e-2 for all."""
import numpy as np
import pytest
import torch

import hns_oracle as O
from hns_amd import config

TAGS = ["free_a3c8", "free_a6c16"]


def _cfg(E, A, C, max_len):
    # (fixtures generated with `done = False` on every stepped tensordict: no reset_pid through the step, see test_oracle_episode._cfg)
    return config.make_cfg({"num_agents": A, "pid_reset": "on_reset", "cylinder": {"max_num": C, "obs_max_cylinder": 3, "min_num": min(4, C)},
                            "env": {"num_envs": E, "max_episode_length": max_len}})


def _s0(arrs, g):
    arrs["cylinders"][:] = g["init_cyl"]
    arrs["drone_state"][..., 0:3], arrs["drone_state"][..., 3:7], arrs["drone_state"][..., 7:13] = g["init_pos"], g["init_rot"], g["init_vel"]
    arrs["target_pos"][:] = g["init_tpos"][:, 0]
    arrs["throttle"][:], arrs["prev_action"][:], arrs["progress"][:] = g["init_throttle"], g["init_prev_action"], g["init_progress"]
    arrs["stats"][:] = g["init_stats"].T
    arrs["pid_integ"][:] = 0
    arrs["pid_last_rate"][..., :3] = 0


def _check(g, t, st, report):
    ds = st["drone_state"]
    d = {"pos": np.abs(ds[..., 0:3] - g["pos"][t]).max(), "rot": np.abs(ds[..., 3:7] - g["rot"][t]).max(),
         "vel": np.abs(ds[..., 7:10] - g["vel"][t][..., :3]).max()}
    e_tpos = np.abs(st["target_pos"] - g["tpos"][t][:, 0]).max(-1)                       # per env
    e_rew = np.abs(st["reward"] - g["reward"][t][..., 0]).max(-1)
    d["tpos"], d["reward"] = e_tpos.max(), e_rew.max()
    report.append(d)
    for k in ("pos", "rot", "vel"):
        assert d[k] < 1e-5, f"step {t}: pursuer {k} drifted {d[k]:.2e}"
    if t < 10:
        assert d["tpos"] < 1e-5 and d["reward"] < 2e-5, f"step {t}: evader position {d['tpos']:.2e}, reward {d['reward']:.2e}"
    assert e_tpos.max() < 5e-2 and e_rew.max() < 5e-2, f"step {t}: evader position {e_tpos.max():.2e}, reward {e_rew.max():.2e}"
    if t == 99:
        assert (e_tpos < 1e-5).mean() >= 0.75 and (e_rew < 2e-5).mean() >= 0.75
    assert (st["done"].astype(bool) == g["done"][t][:, 0]).all(), f"step {t}: done mask differs"


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_free_running_100_steps(golden, tag):
    g = golden(f"g_episode_{tag}")
    E, A, C, T, max_len = (int(x) for x in g["meta"])
    assert T == 100
    c = config.resolve_hns_cfg(_cfg(E, A, C, max_len))
    arrs = O.alloc_buffers(c)
    _s0(arrs, g)
    report = []
    for t in range(T):
        O.step(c, arrs, g["action"][t])
        _check(g, t, arrs, report)
    assert g["done"].any() and report[9]["tpos"] < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_free_running_100_steps(golden, tag):
    from hns_amd.env import HideAndSeek
    g = golden(f"g_episode_{tag}")
    E, A, C, T, max_len = (int(x) for x in g["meta"])
    env = HideAndSeek(_cfg(E, A, C, max_len), headless=True, write_critic_state=True)
    env.reset()
    st = env.export_state()
    _s0(st, g)
    env.import_state(st)                                   # S_0; the line-of-sight column is recomputed from it
    report = []
    for t in range(T):
        env.step(env.rand_step_input(torch.as_tensor(g["action"][t]).to(env.device)))
        _check(g, t, env.export_state(), report)
    print(f"\n{tag}: drift at steps 10 / 50 / 100 — " + "; ".join(
        f"{k} {report[9][k]:.1e} / {report[49][k]:.1e} / {report[99][k]:.1e}" for k in ("pos", "vel", "rot", "tpos", "reward")))
