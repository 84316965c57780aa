"""Hover (BASELINE config 1): the oracle's Hover step against a closed-loop golden
episode of the reference's Hover methods (integrator = the build's spec), teacher forced."""
import numpy as np

import hns_oracle as O
from hns_amd import abi, config


def _cfgs(E, max_len):
    cfg = config.make_hover_cfg({"env": {"num_envs": E, "max_episode_length": max_len}})
    return config.resolve_hover_cfg(cfg)


def load(arrs, g, t):
    if t < 0:
        pos, rot, vel, thr, prog, stats, acc = (g["init_" + k] for k in ("pos", "rot", "vel", "throttle", "progress", "stats", "acc"))
        prev = np.zeros((pos.shape[0], 1, 4), np.float32)
        integ = last = np.zeros(pos.shape, np.float32)
    else:
        pos, rot, vel, thr, prog, stats, acc = (g[k][t] for k in ("pos", "rot", "vel", "throttle", "progress", "stats", "acc"))
        prev, integ, last = g["prev_action"][t], g["integ"][t], g["last"][t]
    arrs["drone_state"][..., 0:3], arrs["drone_state"][..., 3:7], arrs["drone_state"][..., 7:13] = pos, rot, vel
    arrs["throttle"][:], arrs["prev_action"][:], arrs["progress"][:] = thr, prev, prog
    arrs["stats"][:], arrs["acc"][:] = stats.T, acc.T
    arrs["pid_integ"][..., :3], arrs["pid_last_rate"][..., :3] = integ, last


def test_hover_step_teacher_forced(golden):
    g = golden("g_hover")
    E, T, max_len = (int(x) for x in g["meta"])
    c, h = _cfgs(E, max_len)
    arrs = O.alloc_hover_buffers(c)
    saw_done = saw_bonus = False
    for t in range(T):
        load(arrs, g, t - 1)
        O.hover_step(c, h, arrs, g["action"][t])
        kw = dict(rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(arrs["drone_state"][..., 0:3], g["pos"][t], **kw)
        np.testing.assert_allclose(arrs["drone_state"][..., 3:7], g["rot"][t], **kw)
        np.testing.assert_allclose(arrs["drone_state"][..., 7:10], g["vel"][t][..., :3], **kw)
        np.testing.assert_allclose(arrs["obs"], g["obs"][t], **kw)
        np.testing.assert_allclose(arrs["reward"], g["reward"][t][..., 0], rtol=1e-5, atol=2e-6)
        assert (arrs["done"].astype(bool) == g["done"][t][:, 0]).all()
        ref = g["stats"][t].T
        for i, name in enumerate(abi.HOVER_STAT_NAMES):
            # accelerations / jerks divide differences of norms by dt (dt^2): compare relative to their scale
            tol = 2e-2 if "jerk" in name else (2e-3 if "_a_" in name or "acc" in name else 2e-4)
            np.testing.assert_allclose(arrs["stats"][i], ref[i], rtol=1e-4, atol=tol, err_msg=f"{name} step {t}")
        saw_done |= bool(arrs["done"].any())
        saw_bonus |= bool((arrs["stats"][abi.HOVER_STAT_NAMES.index("pos_bonus")] > 0).any())
    assert saw_done
    assert g["init_obs"].shape == (E, 1, 20)


def test_hover_reset_invariants():
    c, h = _cfgs(32, 50)
    arrs = O.alloc_hover_buffers(c)
    arrs["stats"][:] = 3.0
    O.hover_reset(c, h, arrs, None, 11, 0)
    p = arrs["drone_state"][:, 0, :3]
    assert (p[:, :2] >= -1).all() and (p[:, :2] <= 1).all() and (p[:, 2] >= 0.05).all() and (p[:, 2] <= 2.0).all()
    np.testing.assert_allclose(np.linalg.norm(arrs["drone_state"][:, 0, 3:7], axis=-1), 1.0, atol=1e-6)
    assert not arrs["stats"].any() and not arrs["progress"].any()
    np.testing.assert_allclose(arrs["obs"][:, 0, :3], np.array([0, 0, 1], np.float32) - p, atol=1e-7)
    assert (arrs["throttle"] == np.float32(c.hover_throttle)).all()
