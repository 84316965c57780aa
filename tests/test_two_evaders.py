"""Two-evader extension (task.num_targets: 2 — BASELINE config 5 "6-pursuer/2-evader"; NOT in the reference,
spec in include/hns.h).  CPU: the oracle's NT=2 path against its own NT=1 path (an evader that is far away
must not change anything the first evader produces), symmetry under swapping the evaders, capture by either.
GPU (marked): HIP == oracle bit for bit."""
import numpy as np
import pytest
import torch

import hns_oracle as O
from hns_amd import abi, config


def _cfgs(E, A, Cn, **kw):
    base = {"num_agents": A, "drone_detect_radius": 2.0, "cylinder": {"max_num": Cn, "min_num": min(3, Cn)},
            "env": {"num_envs": E, "max_episode_length": 30}}
    base.update(kw)
    c1 = config.resolve_hns_cfg(config.make_cfg(dict(base)))
    c2 = config.resolve_hns_cfg(config.make_cfg(dict(base, num_targets=2)))
    return c1, c2


def test_shapes_and_config_errors():
    c1, c2 = _cfgs(8, 3, 5)
    assert c1.num_targets == 1 and c2.num_targets == 2
    a2 = O.alloc_buffers(c2)
    assert a2["target_pos"].shape == (8, 2, 3) and a2["obs_self"].shape == (8, 3, 24) and a2["state_drones"].shape == (8, 3, 24)
    with pytest.raises(ValueError):
        config.resolve_hns_cfg(config.make_cfg({"num_targets": 3}))
    c3 = config.resolve_hns_cfg(config.make_cfg({"num_targets": 2}, algo={"use_TP_net": 1}))     # the predictor runs once per evader
    assert c3.num_targets == 2
    shp = abi.tp_buffer_shapes(8, 3, 10, 5, 16, num_targets=2)
    assert shp["history"][0] == (16, 10, 16) and shp["pred"][0] == (16, 5, 3) and shp["obs_self"][0] == (8, 3, 24 + 30) and shp["tp_done"][0] == (16,)


def test_far_second_evader_leaves_the_first_untouched():
    E, A, Cn = 96, 3, 8
    c1, c2 = _cfgs(E, A, Cn)
    a1, a2 = O.alloc_buffers(c1), O.alloc_buffers(c2)
    O.reset(c1, a1, None, 5, 0)
    for k in a1:                                            # same start, evader 1 parked far outside every radius
        if k in ("target_pos", "target_vel"):
            a2[k][:, 0] = a1[k]
        elif k in ("obs_self", "state_drones"):
            a2[k][..., :20] = a1[k]
        else:
            a2[k][...] = a1[k]
    a2["target_pos"][:, 1] = (50.0, 50.0, 0.6)
    rng = np.random.default_rng(0)
    for t in range(25):
        act = rng.standard_normal((E, A, 4)).astype(np.float32)
        O.step(c1, a1, act)
        O.step(c2, a2, act)
        for k in ("drone_state", "throttle", "pid_integ", "prev_action", "obs_others", "obs_cylinders", "reward", "done", "progress", "action_error"):
            assert np.array_equal(a1[k], a2[k], equal_nan=True), (k, t)
        assert np.array_equal(a1["target_pos"], a2["target_pos"][:, 0]) and np.array_equal(a1["target_vel"], a2["target_vel"][:, 0])
        assert np.array_equal(a1["obs_self"], a2["obs_self"][..., :20]) and np.array_equal(a1["state_drones"], a2["state_drones"][..., :20])
        assert (a2["obs_self"][..., 20:23] == c2.mask_value).all() and (a2["obs_self"][..., 23] == 0).all()    # never detected
        assert np.array_equal(a2["detect"] & 1, a1["detect"]) and ((a2["detect"] >> 1) == 0).all()
        skip = {abi.STAT_NAMES.index("out_of_arena"), abi.STAT_NAMES.index("blocked")}
        for i in range(abi.HNS_NUM_STATS):
            if i not in skip:
                assert np.array_equal(a1["stats"][i], a2["stats"][i]), abi.STAT_NAMES[i]
    assert (a2["stats"][abi.STAT_NAMES.index("out_of_arena")] == 1).all()              # the parked evader is outside


def test_swapping_the_evaders_swaps_the_outputs():
    E, A, Cn = 64, 4, 6
    _, c2 = _cfgs(E, A, Cn)
    a = O.alloc_buffers(c2)
    O.reset(c2, a, None, 9, 0)
    assert not np.array_equal(a["target_pos"][:, 0], a["target_pos"][:, 1])
    b = {k: v.copy() for k, v in a.items()}
    b["target_pos"] = np.ascontiguousarray(a["target_pos"][:, ::-1])
    rng = np.random.default_rng(1)
    for t in range(20):
        act = rng.standard_normal((E, A, 4)).astype(np.float32)
        O.step(c2, a, act)
        O.step(c2, b, act)
        assert np.array_equal(a["target_pos"], b["target_pos"][:, ::-1]) and np.array_equal(a["target_vel"], b["target_vel"][:, ::-1])
        assert np.array_equal(a["obs_self"][..., 0:3], b["obs_self"][..., 20:23]) and np.array_equal(a["obs_self"][..., 20:23], b["obs_self"][..., 0:3])
        assert np.array_equal(a["obs_self"][..., 3:20], b["obs_self"][..., 3:20])
        for k in ("reward", "drone_state", "stats", "done"):
            assert np.array_equal(a[k], b[k], equal_nan=True), k
        assert np.array_equal(a["detect"] & 1, b["detect"] >> 1) and np.array_equal(a["detect"] >> 1, b["detect"] & 1)


def test_capture_of_either_evader_counts():
    E, A, Cn = 4, 3, 3
    _, c2 = _cfgs(E, A, Cn)
    a = O.alloc_buffers(c2)
    O.reset(c2, a, None, 2, 0)
    a["cylinders"][..., 2] = -20.0                                  # no cylinders: nothing blocks
    a["target_pos"][:, 0] = (-0.5, -0.5, 0.6)
    a["target_pos"][:, 1] = (-0.5, 0.5, 0.6)
    for j in range(A):
        a["drone_state"][:, j, 0:3] = (0.5, 0.3 * (j - 1), 0.6)
    a["drone_state"][1, 0, 0:3] = (-0.5, -0.45, 0.6)               # env 1: next to evader 0
    a["drone_state"][2, 2, 0:3] = (-0.5, 0.45, 0.6)                # env 2: next to evader 1
    O.step(c2, a, np.zeros((E, A, 4), np.float32))
    suc = a["stats"][abi.STAT_NAMES.index("success")]
    assert suc.tolist() == [0.0, 1.0, 1.0, 0.0]
    assert (a["reward"][1] > 15).all() and (a["reward"][2] > 15).all() and (a["reward"][0] < 0).all()   # catch reward is shared


@pytest.mark.gpu
@pytest.mark.parametrize("E,A,Cn", [(100, 3, 8), (64, 6, 16), (257, 1, 3), (130, 2, 5), (16384, 6, 16), (65536, 6, 16)])
def test_hip_two_evaders_matches_oracle(E, A, Cn):
    """(65 536, 6, 16) is BASELINE config 5's per-GPU shard at its stated size: every buffer bit for bit."""
    from hns_amd.env import HideAndSeek
    O.set_threads(32 if E > 30000 else 8 if E > 4096 else 1)
    cfg = config.make_cfg({"num_agents": A, "num_targets": 2, "cylinder": {"max_num": Cn, "min_num": min(3, Cn)},
                           "env": {"num_envs": E, "max_episode_length": 12}}, algo={"critic_input": "state"})
    env = HideAndSeek(cfg)
    env.set_seed(7)
    td = env.reset()
    assert td[("agents", "observation", "state_self")].shape == (E, A, 1, 24)
    host = O.alloc_buffers(env.hcfg)
    O.reset(env.hcfg, host, None, env.seed, env.reset_epoch - 1)
    dev = env.export_state()
    for k in host:
        assert np.array_equal(host[k], dev[k], equal_nan=True), f"reset: {k}"
    g = torch.Generator(device="cpu").manual_seed(E)
    for t in range(14 if E > 30000 else 30):
        act = torch.randn(E, A, 4, generator=g)
        td = env.step(env.rand_step_input(act.to(env.device)))
        O.step(env.hcfg, host, act.numpy())
        if (t + 1) % 12 == 0:
            mask = host["done"].copy()
            assert mask.all()
            rtd = env.rand_step_input()
            rtd.set("_reset", td[("next", "done")].squeeze(-1))
            env.reset(rtd)
            O.reset(env.hcfg, host, mask, env.seed, env.reset_epoch - 1)
        dev = env.export_state()
        for k in host:
            assert np.array_equal(host[k], dev[k], equal_nan=True), f"step {t}: {k}"
    assert (host["detect"] > 1).any() or E < 128                     # the second evader does get detected somewhere


@pytest.mark.gpu
def test_hip_two_evaders_lazy_state_and_bench_shape():
    from hns_amd.env import HideAndSeek
    cfg = config.make_cfg({"num_agents": 6, "num_targets": 2, "cylinder": {"max_num": 16, "min_num": 16}, "env": {"num_envs": 4096}})
    env = HideAndSeek(cfg)                                           # critic_input: obs -> state assembled lazily
    env.reset()
    td = env.step(env.rand_step_input())
    sd = td["next"][("agents", "state")]["state_drones"]
    b = env._bufs
    assert sd.shape == (4096, 6, 24)
    assert torch.equal(sd[..., 0:3], b["drone_state"][..., 0:3] - b["target_pos"][:, None, 0])
    assert torch.equal(sd[..., 20:23], b["drone_state"][..., 0:3] - b["target_pos"][:, None, 1])
    assert torch.equal(sd[..., 3:20], b["obs_self"][..., 3:20])


@pytest.mark.gpu
def test_two_evaders_full_shard_properties_and_rejections():
    """Config 5's shard (6v2, 16 cylinders, 65 536 envs): size-independent properties over 40 steps, determinism, and the
    ."""
    import ctypes as C
    from hns_amd.env import HideAndSeek, HnsError
    E, A, Cn = 65536, 6, 16
    def mk():
        cfg = config.make_cfg({"num_agents": A, "num_targets": 2, "cylinder": {"max_num": Cn, "min_num": Cn}, "env": {"num_envs": E, "max_episode_length": 800}})
        e = HideAndSeek(cfg)
        e.set_seed(11)
        e.reset()
        return e
    e1, e2 = mk(), mk()
    g = torch.Generator(device=e1.device).manual_seed(5)
    for t in range(40):
        act = torch.randn(E, A, 4, generator=g, device=e1.device)
        e1.step(e1.rand_step_input(act)); e2.step(e2.rand_step_input(act))
    a, b = e1.export_state(), e2.export_state()
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), f"not deterministic: {k}"
    assert e1.check_finite()
    q = a["drone_state"][..., 3:7]
    assert np.abs(np.linalg.norm(q, axis=-1) - 1.0).max() < 1e-5                   # unit quaternions
    assert np.linalg.norm(a["drone_state"][..., 7:10], axis=-1).max() <= 1.0 + 1e-6   # |v| <= v_drone
    assert (a["progress"] == 40).all() and not a["done"].any()
    assert np.array_equal(a["obs_self"][..., 3:7], q) and (a["obs_self"][..., 23] == 0).all()
    tp = a["target_pos"]
    assert tp.shape == (E, 2, 3) and np.isfinite(tp).all()


@pytest.mark.gpu
def test_two_evaders_with_the_task_generator():
    """HideAndSeek_envgen with two evaders: task vectors [pursuers | evader 0 | evader 1 | cylinder slots]; every env sits on its task,
    the task reset and the perturbation equal the oracle's, the history fills and is trimmed."""
    import ctypes as C
    from hns_amd.envgen import HideAndSeek_envgen
    E, L, A, Cn = 6144, 4, 3, 6
    cfg = config.make_cfg({"name": "HideAndSeek_envgen", "reset_extra_step": 0, "num_agents": A, "num_targets": 2, "ratio_unif": 0.3, "eval_iter": 1, "R_min": 0.0, "R_max": 1.0,
                           "use_particle_generator": 1, "expand_cylinders": 1, "cylinder": {"max_num": Cn, "min_num": 3},
                           "env": {"num_envs": E, "max_episode_length": L}})
    env = HideAndSeek_envgen(cfg)
    env.set_seed(2)
    env.reset()
    assert env.task_dim == 3 * A + 6 + 3 * Cn and env.all_tasks.shape == (E, env.task_dim)
    g = torch.Generator(device=env.device).manual_seed(0)
    for ep in range(3):
        for t in range(L):
            td = env.step(env.rand_step_input(torch.randn(E, A, 4, generator=g, device=env.device)))
        rtd = env.rand_step_input()
        rtd.set("_reset", td[("next", "done")].squeeze(-1))
        epoch = env.reset_epoch
        env.reset(rtd)
        st = env.export_state()
        placed = np.concatenate([st["drone_state"][..., :3].reshape(E, -1), st["target_pos"].reshape(E, -1), st["cylinders"].reshape(E, -1)], axis=1)
        np.testing.assert_array_equal(placed, env.all_tasks)
        host = O.alloc_buffers(env.hcfg)
        O.reset_tasks(env.hcfg, host, None, env.seed, epoch, env.all_tasks, env.num_unif)
        for k in ("drone_state", "target_pos", "cylinders", "obs_self", "obs_others", "obs_cylinders", "throttle"):
            np.testing.assert_array_equal(host[k][env.num_unif:], st[k][env.num_unif:], err_msg=k)       # the envs placed from perturbed history tasks
    assert len(env.gen_buffer) == 5000                                             # 3 x 6144 kept tasks, trimmed by hns_fps (45 coordinates: the chip-wide kernel)
    hist = env.gen_buffer._history
    out = torch.zeros(500, env.task_dim, device=env.device)
    assert env._lib.hns_perturb_tasks(env._env, hist.data_ptr(), 5000, out.data_ptr(), 500, 1, C.c_float(0.1), C.c_uint64(7), env._stream()) == 0
    ref = O.perturb_tasks(env.hcfg, hist.cpu().numpy(), 500, 1, 0.1, seed=7)
    assert np.array_equal(out.cpu().numpy(), ref)
    sane = O.tasks_sane(env.hcfg, ref) if hasattr(O, "tasks_sane") else None
    assert sane is None or sane.mean() > 0.7                                       # the fallback (a stored task as it is — uniform placements may share a cell) is the exception


def _tp_arrays(c, T, F, rng, scale=2.0):
    NT = 2 if c.num_targets == 2 else 1
    I = abi.tp_frame_dim(c.num_agents, c.num_cylinders, c.tp_use_obstacles)
    t = {k: np.zeros(shape, dtype=dt) for k, (shape, dt) in abi.tp_buffer_shapes(c.num_envs, c.num_agents, T, F, I, NT).items()}
    for k in abi.TP_WEIGHT_FIELDS:
        t[k] = (rng.standard_normal(t[k].shape) * scale / 8.0).astype(np.float32)
    return t


def test_predictor_runs_once_per_evader():
    """Two-evader predictor (extension): unit 2 e + j = the reference's predictor on evader j.  With evader 1 parked out of sight evader 0's
    window, prediction and row part equal the one-evader env's; swapping the evaders swaps the units and the two halves of the rows."""
    E, A, Cn, T, F = 40, 3, 5, 6, 4
    c1, c2 = _cfgs(E, A, Cn)
    a1, a2 = O.alloc_buffers(c1), O.alloc_buffers(c2)
    O.reset(c1, a1, None, 5, 0)
    for k in a1:
        if k in ("target_pos", "target_vel"):
            a2[k][:, 0] = a1[k]
        elif k in ("obs_self", "state_drones"):
            a2[k][..., :20] = a1[k]
        else:
            a2[k][...] = a1[k]
    a2["target_pos"][:, 1] = (50.0, 50.0, 0.6)
    rng = np.random.default_rng(3)
    t1 = _tp_arrays(c1, T, F, rng)
    t2 = _tp_arrays(c2, T, F, rng)
    for k in abi.TP_WEIGHT_FIELDS:
        t2[k] = t1[k].copy()
    R = 3 * F
    for step in range(9):
        if step:
            act = rng.standard_normal((E, A, 4)).astype(np.float32)
            O.step(c1, a1, act)
            O.step(c2, a2, act)
        O.tp_observe(c1, a1, t1, fill=(step == 0))
        O.tp_observe(c2, a2, t2, fill=(step == 0))
        h2, p2 = t2["history"].reshape(E, 2, T, -1), t2["pred"].reshape(E, 2, F, 3)
        assert np.array_equal(h2[:, 0], t1["history"]) and np.array_equal(p2[:, 0], t1["pred"])
        assert np.array_equal(t2["groundtruth"].reshape(E, 2, 3)[:, 0], t1["groundtruth"])
        assert np.array_equal(t2["tp_done"].reshape(E, 2)[:, 0], t1["tp_done"]) and np.array_equal(t2["tp_done"].reshape(E, 2)[:, 1], t1["tp_done"])
        assert (h2[:, 1, -1, 1:7] == c2.mask_value).all()                                   # evader 1 is never detected: masked in its frames
        for k in ("obs_self", "state_drones"):
            assert t2[k].shape == (E, A, 24 + 2 * R)
            assert np.array_equal(t2[k][..., :20 + R], t1[k]), k                          # the reference's row for evader 0 ...
            want = a2[k][..., 20:24] if k == "obs_self" else np.concatenate(                # ... evader 1's relative position (masked / unmasked), 0 ...
                [a2["drone_state"][..., :3] - a2["target_pos"][:, None, 1], np.zeros((E, A, 1), np.float32)], axis=-1)
            assert np.array_equal(t2[k][..., 20 + R:24 + R], want), k
            np.testing.assert_array_equal(t2[k][..., 24 + R:].reshape(E, A, F, 3),        # ... drone - predicted evader 1
                                          a2["drone_state"][:, :, None, :3] - p2[:, None, 1])
    # swap
    _, c = _cfgs(32, 2, 4)
    a = O.alloc_buffers(c)
    O.reset(c, a, None, 7, 0)
    b = {k: v.copy() for k, v in a.items()}
    b["target_pos"] = np.ascontiguousarray(a["target_pos"][:, ::-1])
    b["detect"] = (((a["detect"] & 1) << 1) | (a["detect"] >> 1)).astype(np.uint8)       # what the reset derived from the positions, swapped too
    for k in ("obs_self", "state_drones"):
        b[k][..., 0:3], b[k][..., 20:23] = a[k][..., 20:23], a[k][..., 0:3]
    ta = _tp_arrays(c, T, F, rng)
    tb = {k: v.copy() for k, v in ta.items()}
    for step in range(6):
        if step:
            act = rng.standard_normal((32, 2, 4)).astype(np.float32)
            O.step(c, a, act)
            O.step(c, b, act)
        O.tp_observe(c, a, ta, fill=(step == 0))
        O.tp_observe(c, b, tb, fill=(step == 0))
        assert np.array_equal(ta["pred"].reshape(32, 2, F, 3), tb["pred"].reshape(32, 2, F, 3)[:, ::-1])
        assert np.array_equal(ta["history"].reshape(32, 2, T, -1), tb["history"].reshape(32, 2, T, -1)[:, ::-1])
        assert np.array_equal(ta["obs_self"][..., 3:3 + R], tb["obs_self"][..., 24 + R:]) and np.array_equal(ta["obs_self"][..., 24 + R:], tb["obs_self"][..., 3:3 + R])


@pytest.mark.gpu
@pytest.mark.parametrize("E,A,Cn,obst,T,F", [(200, 3, 5, 0, 10, 5), (130, 6, 16, 0, 10, 5), (64, 2, 8, 1, 4, 8), (257, 4, 12, 1, 7, 3), (1, 1, 3, 0, 10, 5),
                                             (4096, 3, 8, 0, 10, 5)])
def test_hip_two_evader_predictor_matches_oracle(E, A, Cn, obst, T, F):
    """hns_tp_observe with two evaders (one launch over 2 E units, rows of 24 + 6F values) against the oracle: windows, ground truth and
    flags exact, predictions and rows within 1e-5; with steps and masked resets in between."""
    from hns_amd.env import HideAndSeek
    cfg = config.make_cfg({"num_agents": A, "num_targets": 2, "use_obstacles": obst, "cylinder": {"max_num": Cn, "min_num": min(3, Cn)},
                           "env": {"num_envs": E, "max_episode_length": 9}, "history_step": T, "future_predcition_step": F, "drone_detect_radius": 0.9},
                          algo={"use_TP_net": 1, "critic_input": "state"})
    O.set_threads(8 if E > 1024 else 1)
    env = HideAndSeek(cfg)
    env.set_seed(E + A)
    torch.manual_seed(3)
    with torch.no_grad():
        for prm in env.TP.parameters():
            prm.mul_(2.5)
    env.reset()
    tpa = {k: v.cpu().numpy().copy() for k, v in env._tp_bufs.items() if k != "packed"}
    tpa["packed"] = np.zeros(16, np.uint8)
    sd = env.TP.state_dict()
    for f, key in abi.TP_STATE_DICT_KEYS.items():
        tpa[f] = sd[key].detach().cpu().numpy().copy()
    tpa["history"][:] = 0
    O.tp_observe(env.hcfg, env.export_state(), tpa, fill=True)
    assert env._tp_bufs["obs_self"].shape == (E, A, 24 + 6 * F) and env._tp_bufs["history"].shape[0] == 2 * E
    for t in range(12):
        dev = {k: v.cpu().numpy() for k, v in env._tp_bufs.items()}
        assert np.array_equal(dev["history"], tpa["history"]), t
        assert np.array_equal(dev["groundtruth"], tpa["groundtruth"]) and np.array_equal(dev["tp_done"], tpa["tp_done"])
        for k in ("pred", "obs_self", "state_drones"):
            np.testing.assert_allclose(dev[k], tpa[k], rtol=0, atol=1e-5, err_msg=f"{k} at call {t}")
        td = env.step(env.rand_step_input(torch.randn(E, A, 4, device=env.device)))
        assert td[("next", "agents", "observation", "state_self")].shape == (E, A, 1, 24 + 6 * F)
        assert td[("next", "agents", "TP", "TP_input")].shape == (E, 2, T, env.tp_frame_dim)
        O.tp_observe(env.hcfg, env.export_state(), tpa, fill=False)
        done = env._bufs["done"].bool()
        if bool(done.any()) and t % 2 == 0:
            r = env.rand_step_input()
            r.set("_reset", done.clone())
            env.reset(r)
            O.tp_observe(env.hcfg, env.export_state(), tpa, fill=False)
    assert np.abs(tpa["pred"]).max() > 0.05
    # the lazily assembled critic state (write_critic_state off) equals the kernel's rows
    env2 = HideAndSeek(config.make_cfg(dict(cfg.task), algo={"use_TP_net": 1}))
    env2.set_seed(1)
    env2.reset()
    lazy = env2._lazy_state_drones().clone()
    R = 3 * F
    b = env2._bufs
    rp = b["drone_state"][..., None, 0:3] - b["target_pos"].unsqueeze(1)
    assert torch.equal(lazy[..., 0:3], rp[:, :, 0]) and torch.equal(lazy[..., R + 20:R + 23], rp[:, :, 1])
    assert torch.equal(lazy[..., 3:R + 20], env2._tp_bufs["obs_self"][..., 3:R + 20]) and torch.equal(lazy[..., R + 23:], env2._tp_bufs["obs_self"][..., R + 23:])
