"""`task.action_input: motor` — the reference's own task file (`action_transform: PIDrate`, cfg/task/HideAndSeek.yaml:16) with the reference's own
script: scripts/train.py:165-171 puts `PIDRateController` in front of the env, `_inv_call` (utils/torchrl/transforms.py:425-459) replaces
("agents","action") by four rotor commands and leaves ("info","prev_action") / ("stats","action_error_order1") on the stepped tensordict, and the env
starts at `_pre_sim_step` (hideandseek.py:725-744).  Round 5 ran the controller twice in that wiring.

Pins:
  * the rule that decides it (config.resolve_action_input) — from the same key train.py decides on;
  * against the REFERENCE: g_episode_resetpid_a3c5 holds the commands the reference's own `_inv_call` produced (`cmds`), its `prev_action` and
    action error; motor mode fed with them, teacher forced, lands on the golden's states / observation / reward / statistics (1e-5);
  * motor mode == policy mode == oracle, bit for bit, over episodes with reset_pid pulses and masked resets, when the commands come from the
    oracle's restatement of the controller (hns_oracle_ctbr_pid, itself pinned by g_pid);
  * the caller's view: a stand-in transform that does what `_inv_call` does inside TransformedEnv + SyncDataCollector (tests/fake_torchrl/run_collector.py motor).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import hns_oracle as O
from hns_amd import abi, config
from test_reset_pid import KW, TAG, _cfg, _load_oracle, _schedule, _state

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- the rule ---------------------------------------------------------------------------------------------------------------------------------
def test_action_input_follows_the_key_train_py_decides_on(tmp_path):
    # the reference's task file as hydra / load_cfg hands it over: no `action_input`, `action_transform: PIDrate` -> train.py adds the controller -> motor
    y = tmp_path / "HideAndSeek.yaml"
    y.write_text("name: HideAndSeek\naction_transform: PIDrate\nnum_agents: 3\nenv:\n  num_envs: 128\n")
    cfg = config.load_cfg(str(y))
    assert "action_input" not in cfg.task and config.resolve_action_input(cfg.task) == "motor"
    assert config.resolve_hns_cfg(cfg).action_input == abi.HNS_ACTION_MOTOR
    # this build's task file: action_transform none -> nothing in front -> the raw policy output, controller fused into the step
    cfg = config.load_cfg(os.path.join(ROOT, "cfg", "task", "HideAndSeek_hip.yaml"))
    assert config.resolve_action_input(cfg.task) == "policy" and config.resolve_hns_cfg(cfg).action_input == abi.HNS_ACTION_POLICY
    for tr in ("none", "None", None):
        assert config.resolve_action_input({"action_transform": tr}) == "policy"
    for tr in ("PIDrate", "rate", "attitude", "velocity"):          # every controller transform of train.py:148-171 ends in rotor commands
        assert config.resolve_action_input({"action_transform": tr}) == "motor"
    # an explicit key wins either way
    assert config.resolve_action_input({"action_transform": "PIDrate", "action_input": "policy"}) == "policy"
    assert config.resolve_action_input({"action_transform": "none", "action_input": "motor"}) == "motor"
    with pytest.raises(ValueError):
        config.resolve_action_input({"action_input": "ctbr"})
    # make_cfg — this build's programmatic constructor (tests, tools, bench.py) — says "policy" out loud
    cfg = config.make_cfg({})
    assert cfg.task.action_input == "policy" and config.resolve_hns_cfg(cfg).action_input == abi.HNS_ACTION_POLICY
    assert config.resolve_hns_cfg(config.make_cfg({"action_input": "motor"})).action_input == abi.HNS_ACTION_MOTOR
    if os.path.isfile("/root/reference/cfg/task/HideAndSeek.yaml"):  # authoring container only: the reference's file itself, verbatim
        cfg = config.load_cfg("/root/reference/cfg/task/HideAndSeek.yaml")
        assert str(cfg.task.action_transform) == "PIDrate" and config.resolve_hns_cfg(cfg).action_input == abi.HNS_ACTION_MOTOR


# ---- against the reference: the commands of its own _inv_call --------------------------------------------------------------------------------
def _motor_cfg(E, A, C, max_len):
    cfg = _cfg(E, A, C, max_len)
    cfg.task.action_input = "motor"
    return cfg


def _check_motor_step(g, t, st, what):
    """What the env computes from the commands — everything of test_reset_pid._check_step except the controller's own state and outputs
    (pid_integ / pid_last_rate / prev_action / action_error: inputs or untouched in this mode)."""
    ds = st["drone_state"]
    np.testing.assert_allclose(ds[..., 0:3], g["pos"][t], err_msg=what, **KW)
    np.testing.assert_allclose(ds[..., 3:7], g["rot"][t], err_msg=what, **KW)
    np.testing.assert_allclose(ds[..., 7:10], g["vel"][t][..., :3], err_msg=what, **KW)
    np.testing.assert_allclose(ds[..., 10:13], g["vel"][t][..., 3:], rtol=1e-5, atol=3e-5, err_msg=what)
    np.testing.assert_allclose(st["target_pos"], g["tpos"][t][:, 0], err_msg=what, **KW)
    np.testing.assert_allclose(st["throttle"], g["throttle"][t], err_msg=what, **KW)
    np.testing.assert_allclose(st["progress"], g["progress"][t], err_msg=what)
    np.testing.assert_allclose(st["obs_self"], g["state_self"][t][:, :, 0], err_msg=what, **KW)
    np.testing.assert_allclose(st["obs_others"], g["state_others"][t], err_msg=what, **KW)
    np.testing.assert_allclose(st["obs_cylinders"], g["cylinders"][t], err_msg=what, **KW)
    np.testing.assert_allclose(st["state_drones"], g["state_drones"][t], err_msg=what, **KW)
    np.testing.assert_allclose(st["reward"], g["reward"][t][..., 0], rtol=1e-5, atol=1e-6, err_msg=what)
    assert (st["done"].astype(bool) == g["done"][t][:, 0]).all(), what
    ref = g["stats"][t].T
    for i, name in enumerate(abi.STAT_NAMES):                          # incl. action_error_order1_mean / _max: fed from the transform's key (:731-733)
        np.testing.assert_allclose(st["stats"][i], ref[i], rtol=1e-5, atol=3e-6, err_msg=f"{what}: {name}")
    # inputs stay what the caller wrote
    assert np.array_equal(st["prev_action"], g["prev_action"][t]) and np.array_equal(st["action_error"], g["aerr"][t]), what


def _golden_motor_steps(g):
    """(t, state to load, what the transform left on the stepped tensordict) for every step of the golden; resets are loaded, not replayed
    (test_reset_pid replays them: the reset path does not depend on the action input)."""
    prev = -1
    for kind, i in _schedule(g):
        if kind == "reset":
            prev = ("reset", i)
            continue
        st, last = _state(g, prev, None)
        yield i, st, last, dict(cmds=g["cmds"][i], prev_action=g["prev_action"][i], aerr=g["aerr"][i])
        prev = i


def test_oracle_motor_mode_lands_on_the_reference_golden(golden):
    g = golden(TAG)
    E, A, C, T, max_len = (int(x) for x in g["meta"])
    c = config.resolve_hns_cfg(_motor_cfg(E, A, C, max_len))
    assert c.action_input == abi.HNS_ACTION_MOTOR
    arrs = O.alloc_buffers(c)
    n = 0
    for t, st, last, keys in _golden_motor_steps(g):
        _load_oracle(arrs, st, last)
        arrs["done"][:] = g["root_done"][t][:, 0]
        arrs["prev_action"][...] = keys["prev_action"]                  # hideandseek.py:729
        arrs["action_error"][...] = keys["aerr"]                        # :731
        integ, lastr = arrs["pid_integ"].copy(), arrs["pid_last_rate"][..., :3].copy()
        O.step(c, arrs, keys["cmds"])
        _check_motor_step(g, t, arrs, f"step {t}")
        assert np.array_equal(arrs["pid_integ"], integ) and np.array_equal(arrs["pid_last_rate"][..., :3], lastr)   # the caller's controller owns them
        n += 1
    assert n == T


# ---- motor == policy == oracle, closed loop --------------------------------------------------------------------------------------------------
class HostController:
    """The caller's side of `action_input: motor` on host arrays: PIDRateController._inv_call (transforms.py:425-459) around the body-rate PID
    (lee_position_controller.py:476-550) through the oracle's restatement of both — the state (`integ`, `last`) lives HERE, as it lives in the
    torch module in the reference."""

    def __init__(self, cfg, E, A):
        self.c, self.E, self.A = cfg, E, A
        self.integ = np.zeros((E * A, 3), np.float32)
        self.last = np.zeros((E * A, 3), np.float32)

    def inv(self, action, drone_state, prev_action, root_done):
        E, A = self.E, self.A
        ds = np.asarray(drone_state, np.float32).reshape(E * A, 13)
        r = O.ctbr_pid(self.c, action, ds[:, 3:7], ds[:, 10:13], np.repeat(np.asarray(root_done).reshape(E).astype(np.uint8), A), prev_action, self.integ, self.last)
        self.integ, self.last = r["integ"], r["last"]
        return dict(cmds=r["cmd"].reshape(E, A, 4), prev_action=r["prev_action"].reshape(E, A, 4), aerr=r["aerr"].reshape(E, A))


SAME_IN_BOTH_MODES = [k for k in abi.BUFFER_FIELDS if k not in ("pid_integ", "pid_last_rate", "ctbr", "target_rate", "reset_pid")]


def _compare_modes(pol, mot, ctl, what):
    for k in SAME_IN_BOTH_MODES:
        if k in pol and pol[k] is not None:
            assert np.array_equal(pol[k], mot[k], equal_nan=True), f"{what}: {k} differs between policy and motor input"
    # the line-of-sight column is the env's in both modes; the controller state sits in the env (policy) or with the caller (motor)
    assert np.array_equal(pol["pid_last_rate"][..., 3], mot["pid_last_rate"][..., 3]), what
    E, A = pol["pid_integ"].shape[:2]
    assert np.array_equal(pol["pid_integ"][..., :3].reshape(E * A, 3), ctl.integ) and np.array_equal(pol["pid_last_rate"][..., :3].reshape(E * A, 3), ctl.last), what


@pytest.mark.parametrize("A,C,NT", [(3, 5, 1), (2, 8, 2)])
def test_oracle_motor_equals_policy_over_episodes(A, C, NT):
    E, L, T = 96, 9, 31
    base = {"num_agents": A, "num_targets": NT, "use_deployment": 1, "init_smoothness_coef": 0.5, "cylinder": {"max_num": C, "min_num": min(3, C)},
            "env": {"num_envs": E, "max_episode_length": L}}
    cp = config.resolve_hns_cfg(config.make_cfg(base))
    cm = config.resolve_hns_cfg(config.make_cfg(dict(base, action_input="motor")))
    pol, mot = O.alloc_buffers(cp), O.alloc_buffers(cm)
    O.reset(cp, pol, None, 3, 0)
    O.reset(cm, mot, None, 3, 0)
    pol["progress"][:] = mot["progress"][:] = np.arange(E) % 4          # episodes end at different steps: reset_pid pulses persist until the masked reset
    ctl = HostController(cm, E, A)
    rng = np.random.default_rng(7)
    epoch, pulses = 1, 0
    for t in range(T):
        a = (rng.standard_normal((E, A, 4)) * 0.7).astype(np.float32)
        root_done = mot["done"].copy()
        pulses += int(root_done.sum())
        keys = ctl.inv(a, mot["drone_state"], mot["prev_action"], root_done)
        mot["prev_action"][...] = keys["prev_action"]
        mot["action_error"][...] = keys["aerr"]
        O.step(cp, pol, a)
        O.step(cm, mot, keys["cmds"])
        _compare_modes(pol, mot, ctl, f"step {t}")
        if t % 5 == 4 and pol["done"].any():                              # the collector's masked reset, a few steps late for some envs
            m = pol["done"].copy()
            O.reset(cp, pol, m, 3, epoch)
            O.reset(cm, mot, m, 3, epoch)
            epoch += 1
            _compare_modes(pol, mot, ctl, f"reset after step {t}")
    assert pulses > E and epoch > 3


@pytest.mark.gpu
def test_hip_motor_mode_lands_on_the_reference_golden(golden):
    """The HIP step fed with the commands of the reference's own `_inv_call`: golden states / observation / reward (1e-5) and the oracle (bit for bit)."""
    import torch
    from hns_amd.env import HideAndSeek
    from hns_amd.tensordict_shim import TensorDict
    g = golden(TAG)
    E, A, C, T, max_len = (int(x) for x in g["meta"])
    env = HideAndSeek(_motor_cfg(E, A, C, max_len))
    assert env.action_input == "motor" and env.step_mapping == "tile"
    env.set_seed(0)
    env.reset()
    c = env.hcfg
    host = env.export_state()
    for t, st, last, keys in _golden_motor_steps(g):
        cur = env.export_state()
        for k, v in st.items():
            cur[k] = v
        cur["pid_last_rate"][..., :3] = last
        cur["done"][:] = g["root_done"][t][:, 0]
        env.import_state(cur)
        for k in host:
            host[k][...] = env.export_state()[k]
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(env.device)      # noqa: E731
        td = TensorDict({"agents": {"action": dev(keys["cmds"])}, "info": {"prev_action": dev(keys["prev_action"])},
                         "stats": {"action_error_order1": dev(keys["aerr"])}, "done": dev(g["root_done"][t])}, env.batch_size)
        out = env.step(td)
        host["prev_action"][...] = keys["prev_action"]
        host["action_error"][...] = keys["aerr"]
        O.step(c, host, keys["cmds"])
        now = env.export_state()
        for k in host:
            assert np.array_equal(host[k], now[k], equal_nan=True), f"step {t}: {k} differs from the oracle"
        _check_motor_step(g, t, now, f"step {t}")
        # the transform's keys stay the caller's: the env hands out `next` only
        assert out[("info", "prev_action")].data_ptr() == td[("info", "prev_action")].data_ptr()
        assert torch.equal(out[("next", "info", "prev_action")], td[("info", "prev_action")])     # hideandseek.py:729: info.prev_action with the next observation


@pytest.mark.gpu
@pytest.mark.parametrize("A,C,NT,E", [(3, 8, 1, 4096), (3, 5, 1, 200), (6, 16, 2, 1024), (2, 6, 1, 320)])
def test_hip_motor_equals_hip_policy_equals_oracle(A, C, NT, E):
    """Closed loop, two HIP envs and the oracle: raw actions into the fused controller (policy) against the same actions through the caller-side
    controller and the motor input — every buffer both modes share is bit-identical, step by step, across reset_pid pulses and masked resets."""
    import torch
    from hns_amd.env import HideAndSeek
    from hns_amd.tensordict_shim import TensorDict
    L, T = 7, 24
    K = 3 if C != 6 else 6                                              # (2, 6): the wide k-nearest instantiation
    base = {"num_agents": A, "num_targets": NT, "use_deployment": 1, "init_smoothness_coef": 0.5,
            "cylinder": {"max_num": C, "min_num": min(3, C), "obs_max_cylinder": K}, "env": {"num_envs": E, "max_episode_length": L}}
    pol = HideAndSeek(config.make_cfg(base), headless=True, write_critic_state=True)
    mot = HideAndSeek(config.make_cfg(dict(base, action_input="motor")), headless=True, write_critic_state=True)
    for e in (pol, mot):
        e.set_seed(11)
        e.reset()
    host = mot.export_state()
    ctl = HostController(mot.hcfg, E, A)
    g = torch.Generator().manual_seed(3)
    for t in range(T):
        a = torch.randn(E, A, 4, generator=g) * 0.7
        cur = mot.export_state()
        keys = ctl.inv(a.numpy(), cur["drone_state"], cur["prev_action"], cur["done"])
        pol.step(pol.rand_step_input(a.to(pol.device)))
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(mot.device)      # noqa: E731
        mot.step(TensorDict({"agents": {"action": dev(keys["cmds"])}, "info": {"prev_action": dev(keys["prev_action"])},
                             "stats": {"action_error_order1": dev(keys["aerr"])}}, mot.batch_size))
        host["prev_action"][...] = keys["prev_action"]
        host["action_error"][...] = keys["aerr"]
        O.step(mot.hcfg, host, keys["cmds"])
        p, m = pol.export_state(), mot.export_state()
        _compare_modes(p, m, ctl, f"step {t}")
        for k in host:
            assert np.array_equal(host[k], m[k], equal_nan=True), f"step {t}: {k} differs from the oracle (motor input)"
        if t % 4 == 3 and p["done"].any():
            mask = torch.from_numpy(p["done"].astype(bool))
            epoch = mot.reset_epoch
            for e in (pol, mot):
                e.reset(TensorDict({"_reset": mask.to(e.device)}, e.batch_size))
            O.reset(mot.hcfg, host, p["done"], mot.seed, epoch)
            _compare_modes(pol.export_state(), mot.export_state(), ctl, f"reset after step {t}")
    assert pol.reset_epoch > 3


@pytest.mark.gpu
def test_hip_motor_mode_without_the_transforms_keys_fails_as_the_reference_does():
    """hideandseek.py:729-731 indexes the stepped tensordict: a caller that did not run the transform gets a KeyError, not a flying-but-wrong env."""
    import torch
    from hns_amd.env import HideAndSeek
    env = HideAndSeek(config.make_cfg({"action_input": "motor", "env": {"num_envs": 64}}))
    env.reset()
    with pytest.raises(KeyError, match="action_input"):
        env.step(env.rand_step_input(torch.zeros(64, 3, 4, device=env.device)))


@pytest.mark.gpu
def test_reference_task_file_through_transformed_env_and_collector():
    """The wiring of scripts/train.py:165-205 with the reference's task file unchanged (`action_transform: PIDrate`): a stand-in transform doing what
    `_inv_call` does sits in TransformedEnv, the env derives `action_input: motor` from the same key; rewards and `done` of four rollouts are
    replayed on the oracle's fused controller from the RAW policy actions (tests/fake_torchrl/run_collector.py motor)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_torchrl", "run_collector.py"), "0", "motor"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info["action_input"] == "motor" and info["rollouts"] == 4 and info["masked_resets"] >= 2


def test_hover_refuses_the_double_controller():
    """The plumbing task has no motor input: with the reference's Hover.yaml (`action_transform: PIDrate`) as hydra hands it over it raises instead of running
    the controller twice; this build's programmatic constructor states `action_input: policy`."""
    config.resolve_hover_cfg(config.make_hover_cfg({}))
    cfg = config.make_hover_cfg({})
    del cfg.task["action_input"]
    with pytest.raises(NotImplementedError, match="action_transform"):
        config.resolve_hover_cfg(cfg)
    cfg.task.action_transform = "none"
    config.resolve_hover_cfg(cfg)


@pytest.mark.gpu
def test_hip_motor_mode_with_the_predictor_and_the_generator():
    """The reference's defaults together — `action_transform: PIDrate` (motor input), `algo.use_TP_net: 1`, the task generator: the predictor's rows, window and
    predictions and the generator's task batches are bit-identical between the motor-input env and the policy-input env fed the same actions through the two paths."""
    import torch
    from hns_amd.envgen import HideAndSeek_envgen
    from hns_amd.tensordict_shim import TensorDict
    E, A, L = 256, 3, 6
    base = {"name": "HideAndSeek_envgen", "num_agents": A, "eval_iter": 1, "R_min": 0.0, "R_max": 1.0, "ratio_unif": 0.3, "use_particle_generator": 1,
            "cylinder": {"max_num": 5, "min_num": 3}, "env": {"num_envs": E, "max_episode_length": L}}
    pol = HideAndSeek_envgen(config.make_cfg(base, algo={"use_TP_net": 1}))
    mot = HideAndSeek_envgen(config.make_cfg(dict(base, action_input="motor"), algo={"use_TP_net": 1}))
    mot.TP.load_state_dict(pol.TP.state_dict())                        # the same predictor on both sides
    for e in (pol, mot):
        e.set_seed(5)
        e.reset()
    ctl = HostController(mot.hcfg, E, A)
    g = torch.Generator().manual_seed(9)
    for t in range(3 * L + 2):
        a = torch.randn(E, A, 4, generator=g) * 0.7
        cur = mot.export_state()
        keys = ctl.inv(a.numpy(), cur["drone_state"], cur["prev_action"], cur["done"])
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(mot.device)      # noqa: E731
        tdp = pol.step(pol.rand_step_input(a.to(pol.device)))
        mot.step(TensorDict({"agents": {"action": dev(keys["cmds"])}, "info": {"prev_action": dev(keys["prev_action"])},
                             "stats": {"action_error_order1": dev(keys["aerr"])}}, mot.batch_size))
        _compare_modes(pol.export_state(), mot.export_state(), ctl, f"step {t}")
        for k in ("obs_self", "pred", "history", "groundtruth", "tp_done"):
            assert torch.equal(pol._tp_bufs[k], mot._tp_bufs[k]), f"step {t}: predictor buffer {k} differs between policy and motor input"
        if (t + 1) % L == 0:                                           # lock-step episode end: both envs reset through the generator
            mask = tdp[("next", "done")].squeeze(-1).clone()
            for e in (pol, mot):
                e.reset(TensorDict({"_reset": mask.to(e.device)}, e.batch_size))
            assert torch.equal(pol._tasks_dev, mot._tasks_dev) and len(pol.gen_buffer) == len(mot.gen_buffer)
            _compare_modes(pol.export_state(), mot.export_state(), ctl, f"generator reset after step {t}")
    assert len(pol.gen_buffer) > 0
