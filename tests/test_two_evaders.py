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
    with pytest.raises(NotImplementedError):
        config.resolve_hns_cfg(config.make_cfg({"num_targets": 2}, algo={"use_TP_net": 1}))


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
    combination the extension does not support fails loudly instead of silently (predictor)."""
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
    # rejection: the predictor's frame holds one evader
    with pytest.raises(NotImplementedError):
        config.make_cfg({"num_agents": A, "num_targets": 2, "cylinder": {"max_num": Cn, "min_num": Cn}, "env": {"num_envs": 64}}, algo={"use_TP_net": 1}) and \
            HideAndSeek(config.make_cfg({"num_agents": A, "num_targets": 2, "cylinder": {"max_num": Cn, "min_num": Cn}, "env": {"num_envs": 64}}, algo={"use_TP_net": 1}))


@pytest.mark.gpu
def test_two_evaders_with_the_task_generator():
    """HideAndSeek_envgen with two evaders: task vectors [pursuers | evader 0 | evader 1 | cylinder slots]; every env sits on its task,
    the task reset and the perturbation equal the oracle's, the history fills and is trimmed."""
    import ctypes as C
    from hns_amd.envgen import HideAndSeek_envgen
    E, L, A, Cn = 6144, 4, 3, 6
    cfg = config.make_cfg({"name": "HideAndSeek_envgen", "num_agents": A, "num_targets": 2, "ratio_unif": 0.3, "eval_iter": 1, "R_min": 0.0, "R_max": 1.0,
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
