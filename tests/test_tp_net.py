"""SURVEY §8 N2 — TP_net inside the observation: oracle observation (20-dim rows, broadcast_detect)
+ hns_amd.tp_net assembly against the reference's `_compute_state_and_obs` with use_TP_net=1
(golden g_tp_obs: 14 consecutive calls, so the 10-frame history is exercised)."""
import numpy as np
import torch

import hns_oracle as O
from hns_amd import config
from hns_amd.tp_net import TPNet
from tp_reference import TPObservation


def _load_weights(tp, g):
    sd = {k: torch.from_numpy(g["w_" + k.replace(".", "_")]) for k in tp.state_dict()}
    tp.load_state_dict(sd)


import pytest

GOLDENS = ["g_tp_obs", "g_tp_obs_a6", "g_tp_obs_obst", "g_tp_obs_obst_c8"]     # 3 pursuers, 6 pursuers, 3 pursuers + task.use_obstacles (5 / 8 cylinders: 31 / 40 values)


def _golden_cfg(g):
    E, A, C, T, max_len = (int(x) for x in g["meta"])
    obst = int(g["use_obstacles"]) if "use_obstacles" in g else 0
    cfg = config.make_cfg({"num_agents": A, "drone_detect_radius": 0.9, "use_obstacles": obst, "cylinder": {"max_num": C, "min_num": 4},
                           "env": {"num_envs": E, "max_episode_length": max_len}}, algo={"use_TP_net": 1})
    return cfg, obst


@pytest.mark.parametrize("name", GOLDENS)
def test_tp_observation_matches_reference(golden, name):
    g = golden(name)
    E, A, C, T, max_len = (int(x) for x in g["meta"])
    cfg, obst = _golden_cfg(g)
    c = config.resolve_hns_cfg(cfg)
    tp = TPNet(7 + 3 * A + (3 * C if obst else 0), 15, 5, 1)
    _load_weights(tp, g)
    helper = TPObservation(tp, A, 0.9, 1.2, max_len, cylinder_size=float(c.cylinder_size) if obst else None)
    arrs = O.alloc_buffers(c)
    arrs["cylinders"][:] = g["cyl"]
    saw_masked = False
    for t in range(T):
        arrs["drone_state"][..., 0:3], arrs["drone_state"][..., 3:7], arrs["drone_state"][..., 7:13] = g["pos"][t], g["rot"][t], g["vel"][t]
        arrs["throttle"][:] = g["throttle"][t]
        arrs["target_pos"][:] = g["tpos"][t][:, 0]
        arrs["progress"][:] = g["progress"][t]
        _, bdet, _ = O.obs_reward(c, arrs)
        assert (bdet == g["broadcast_detect"][t][:, 0]).all()
        saw_masked |= bool((~bdet).any())
        ss, sd, tpd = helper(torch.from_numpy(arrs["obs_self"]), torch.from_numpy(g["pos"][t]), torch.from_numpy(g["tpos"][t][:, 0]),
                             torch.from_numpy(g["tvel"][t][:, 0]), torch.from_numpy(g["progress"][t]), torch.from_numpy(bdet),
                             cylinders=torch.from_numpy(g["cyl"]))
        np.testing.assert_allclose(tpd["TP_input"].numpy(), g["TP_input"][t], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ss.numpy(), g["state_self"][t][:, :, 0], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(sd.numpy(), g["state_drones"][t], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(tpd["TP_groundtruth"].numpy(), g["TP_groundtruth"][t], rtol=1e-6, atol=1e-6)
        assert (tpd["TP_done"].numpy() == g["TP_done"][t]).all()
    assert saw_masked and ss.shape == (E, A, 35)


def test_tp_net_state_dict_is_reference_compatible():
    tp = TPNet(16, 15, 5, 1)
    assert set(tp.state_dict()) == {"lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0", "fc.weight", "fc.bias"}
    assert tp(torch.zeros(4, 10, 16)).shape == (4, 15)


@pytest.mark.parametrize("name", GOLDENS)
def test_oracle_tp_observe_matches_reference(golden, name):
    """The C restatement of the whole TP branch (window, LSTM, output layer, rows) against the reference's
    `_compute_state_and_obs` + its own TP_net, for 3 pursuers (16-value frames), 6 (25-value frames) and
    3 with task.use_obstacles (31-value frames: the cylinders ride in the frame)."""
    g = golden(name)
    E, A, C, T, max_len = (int(x) for x in g["meta"])
    cfg, obst = _golden_cfg(g)
    c = config.resolve_hns_cfg(cfg)
    assert int(c.tp_use_obstacles) == obst
    arrs = O.alloc_buffers(c)
    tpa = O.alloc_tp_buffers(c, 10, 5)
    from hns_amd import abi
    for f, key in abi.TP_STATE_DICT_KEYS.items():
        tpa[f][...] = g["w_" + key.replace(".", "_")]
    arrs["cylinders"][:] = g["cyl"]
    for t in range(T):
        arrs["drone_state"][..., 0:3], arrs["drone_state"][..., 3:7], arrs["drone_state"][..., 7:13] = g["pos"][t], g["rot"][t], g["vel"][t]
        arrs["throttle"][:] = g["throttle"][t]
        arrs["target_pos"][:] = g["tpos"][t][:, 0]
        arrs["target_vel"][:] = g["tvel"][t][:, 0]
        arrs["progress"][:] = g["progress"][t]
        _, bdet, _ = O.obs_reward(c, arrs)
        arrs["detect"][:] = bdet
        O.tp_observe(c, arrs, tpa, fill=(t == 0))
        np.testing.assert_allclose(tpa["history"], g["TP_input"][t], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(tpa["obs_self"], g["state_self"][t][:, :, 0], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(tpa["state_drones"], g["state_drones"][t], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(tpa["groundtruth"], g["TP_groundtruth"][t], rtol=1e-6, atol=1e-6)
        assert (tpa["tp_done"].astype(bool) == g["TP_done"][t][:, 0]).all()
