"""`reset_pid = tensordict['done']` (transforms.py:449-454 -> lee_position_controller.py:497-502) and the reset semantics of the
reference (hideandseek.py:609-723: the controller is NOT touched by `_reset_idx`; one extra physics step of the whole scene at its end)
against g_episode_resetpid_a3c5: a closed-loop episode produced by the reference's own `_inv_call` / `_pre_sim_step` / observation /
reward code in which the stepped tensordict carries the real root `done`, crossing two resets of the done envs (make_golden.py
gen_episode_resetpid).  Teacher forced: every step and every reset starts from the golden state, so one-step errors cannot hide.

task.pid_reset = reference, task.reset_extra_step = 1 (this build's defaults since round 4).  CPU: the oracle; GPU: the HIP path through
the C ABI, against the golden (1e-5) AND against the oracle (bit for bit)."""
import numpy as np
import pytest

import hns_oracle as O
from hns_amd import abi, config

TAG = "g_episode_resetpid_a3c5"


def _cfg(E, A, C, max_len):
    # use_eval: the reset attitude box collapses to rpy = 0 (hideandseek.py:283-313), so a reset from a task vector is deterministic
    return config.make_cfg({"num_agents": A, "use_eval": 1, "pid_reset": "reference", "reset_extra_step": 1,
                            "cylinder": {"max_num": C, "obs_max_cylinder": 3, "min_num": min(4, C)},
                            "env": {"num_envs": E, "max_episode_length": max_len}}, algo={"critic_input": "state"})


def _state(g, t, resets):
    """Everything the next call reads, as the golden holds it after step t (t = -1: initial) or after reset event r = resets[t]."""
    E = g["init_pos"].shape[0]
    if isinstance(t, tuple):                                           # ("reset", r)
        r = t[1]
        pre = "reset_"
        pos, rot, vel, tpos = g[pre + "pos"][r], g[pre + "rot"][r], g[pre + "vel"][r], g[pre + "tpos"][r]
        thr, prev, prog, stats = g[pre + "throttle"][r], g[pre + "prev_action"][r], g[pre + "progress"][r], g[pre + "stats"][r]
        integ, last, cyl = g[pre + "integ"][r], g[pre + "last"][r], g[pre + "cyl"][r]
    elif t < 0:
        pos, rot, vel, tpos = g["init_pos"], g["init_rot"], g["init_vel"], g["init_tpos"]
        thr, prev, prog, stats = g["init_throttle"], g["init_prev_action"], g["init_progress"], g["init_stats"]
        integ = last = np.zeros(pos.shape, np.float32)
        cyl = g["init_cyl"]
    else:
        pos, rot, vel, tpos = g["pos"][t], g["rot"][t], g["vel"][t], g["tpos"][t]
        thr, prev, prog, stats = g["throttle"][t], g["prev_action"][t], g["progress"][t], g["stats"][t]
        integ, last = g["integ"][t], g["last"][t]
        cyl = None                                                     # unchanged by a step
    ds = np.concatenate([pos, rot, vel], axis=-1).astype(np.float32)
    out = {"drone_state": ds, "target_pos": tpos[:, 0].astype(np.float32), "throttle": thr.astype(np.float32), "prev_action": prev.astype(np.float32),
           "progress": prog.astype(np.float32), "stats": np.ascontiguousarray(stats.T.astype(np.float32)),
           "pid_integ": np.concatenate([integ, np.zeros((*integ.shape[:-1], 1), np.float32)], -1).astype(np.float32)}
    if cyl is not None:
        out["cylinders"] = cyl.astype(np.float32)
    return out, last.astype(np.float32)


def _load_oracle(arrs, st, last):
    for k, v in st.items():
        arrs[k][...] = v
    arrs["pid_last_rate"][..., :3] = last


def _schedule(g):
    """[(kind, index)]: ("reset", r) before step reset_step[r], then ("step", t)."""
    T = int(g["meta"][3])
    rs = {int(s): r for r, s in enumerate(g["reset_step"])}
    out = []
    for t in range(T):
        if t in rs:
            out.append(("reset", rs[t]))
        out.append(("step", t))
    return out


KW = dict(rtol=1e-5, atol=1e-5)


def _check_step(g, t, st, what):
    ds = st["drone_state"]
    np.testing.assert_allclose(ds[..., 0:3], g["pos"][t], err_msg=what, **KW)
    np.testing.assert_allclose(ds[..., 3:7], g["rot"][t], err_msg=what, **KW)
    np.testing.assert_allclose(ds[..., 7:10], g["vel"][t][..., :3], err_msg=what, **KW)
    np.testing.assert_allclose(ds[..., 10:13], g["vel"][t][..., 3:], rtol=1e-5, atol=3e-5, err_msg=what)
    np.testing.assert_allclose(st["target_pos"], g["tpos"][t][:, 0], err_msg=what, **KW)
    # the controller state is where reset_pid shows: integrator and last body rate restart from zero in the pulsed envs
    np.testing.assert_allclose(st["pid_integ"][..., :3], g["integ"][t], rtol=1e-5, atol=1e-4, err_msg=what)      # deg: |values| up to 33
    np.testing.assert_allclose(st["pid_last_rate"][..., :3], g["last"][t], rtol=1e-5, atol=2e-3, err_msg=what)   # deg/s: |values| up to hundreds
    np.testing.assert_allclose(st["throttle"], g["throttle"][t], err_msg=what, **KW)
    np.testing.assert_allclose(st["prev_action"], g["prev_action"][t], err_msg=what, **KW)
    np.testing.assert_allclose(st["action_error"], g["aerr"][t], err_msg=what, **KW)
    np.testing.assert_allclose(st["progress"], g["progress"][t], err_msg=what)
    np.testing.assert_allclose(st["obs_self"], g["state_self"][t][:, :, 0], err_msg=what, **KW)
    np.testing.assert_allclose(st["obs_others"], g["state_others"][t], err_msg=what, **KW)
    np.testing.assert_allclose(st["obs_cylinders"], g["cylinders"][t], err_msg=what, **KW)
    np.testing.assert_allclose(st["state_drones"], g["state_drones"][t], err_msg=what, **KW)
    np.testing.assert_allclose(st["reward"], g["reward"][t][..., 0], rtol=1e-5, atol=1e-6, err_msg=what)
    assert (st["done"].astype(bool) == g["done"][t][:, 0]).all(), what
    ref = g["stats"][t].T
    for i, name in enumerate(abi.STAT_NAMES):
        np.testing.assert_allclose(st["stats"][i], ref[i], rtol=1e-5, atol=3e-6, err_msg=f"{what}: {name}")


def _check_reset(g, r, st, before, what):
    mask = g["reset_mask"][r].astype(bool)
    ds = st["drone_state"]
    np.testing.assert_allclose(ds[..., 0:3], g["reset_pos"][r], err_msg=what, **KW)
    np.testing.assert_allclose(ds[..., 3:7], g["reset_rot"][r], err_msg=what, **KW)
    np.testing.assert_allclose(ds[..., 7:13], g["reset_vel"][r], rtol=1e-5, atol=3e-5, err_msg=what)
    assert (ds[mask][..., 9] < -0.09).all(), what                     # one step of free fall behind the placement (the extra physics step)
    np.testing.assert_allclose(st["target_pos"], g["reset_tpos"][r][:, 0], err_msg=what, **KW)
    np.testing.assert_allclose(st["cylinders"], g["reset_cyl"][r], err_msg=what, **KW)
    np.testing.assert_allclose(st["throttle"], g["reset_throttle"][r], err_msg=what, **KW)
    np.testing.assert_allclose(st["prev_action"], g["reset_prev_action"][r], rtol=1e-6, atol=1e-6, err_msg=what)
    np.testing.assert_allclose(st["progress"], g["reset_progress"][r], err_msg=what)
    np.testing.assert_allclose(st["stats"], g["reset_stats"][r].T, rtol=1e-6, atol=1e-6, err_msg=what)
    # `_reset_idx` leaves the controller alone: bit for bit what it held
    assert (st["pid_integ"][..., :3] == before["pid_integ"][..., :3]).all() and (st["pid_last_rate"][..., :3] == before["pid_last_rate"][..., :3]).all(), what
    np.testing.assert_allclose(st["pid_integ"][..., :3], g["reset_integ"][r], rtol=1e-5, atol=1e-4, err_msg=what)
    np.testing.assert_allclose(st["obs_self"], g["reset_state_self"][r][:, :, 0], err_msg=what, **KW)
    np.testing.assert_allclose(st["obs_others"], g["reset_state_others"][r], err_msg=what, **KW)
    np.testing.assert_allclose(st["obs_cylinders"], g["reset_cylinders"][r], err_msg=what, **KW)
    np.testing.assert_allclose(st["state_drones"], g["reset_state_drones"][r], err_msg=what, **KW)
    assert not st["done"][mask].any() and st["done"][~mask].astype(bool).tolist() == before["done"][~mask].astype(bool).tolist(), what


def _run(g, step_fn, reset_fn, load_fn, read_fn):
    pulses = 0
    prev = -1
    for kind, i in _schedule(g):
        st, last = _state(g, prev, None)
        if kind == "reset":
            t = int(g["reset_step"][i])
            load_fn(st, last, done=g["done"][t - 1][:, 0], target_vel=g["tvel"][t - 1][:, 0])
            before = read_fn()
            reset_fn(g["reset_mask"][i], g["reset_tasks"][i])
            _check_reset(g, i, read_fn(), before, f"reset {i} before step {t}")
            prev = ("reset", i)
        else:
            rd = g["root_done"][i][:, 0]
            pulses += int(rd.sum())
            load_fn(st, last, done=rd, target_vel=None)
            step_fn(g["action"][i])
            _check_step(g, i, read_fn(), f"step {i}")
            prev = i
    assert pulses > 20            # the sequence does pulse reset_pid, before the first reset and persistently before the second


def test_oracle_reset_pid_across_resets(golden):
    g = golden(TAG)
    E, A, C, T, max_len = (int(x) for x in g["meta"])
    c = config.resolve_hns_cfg(_cfg(E, A, C, max_len))
    assert c.pid_reset_on_reset == 0 and c.reset_extra_step == 1
    arrs = O.alloc_buffers(c)

    def load(st, last, done, target_vel):
        _load_oracle(arrs, st, last)
        arrs["done"][:] = done
        if target_vel is not None:
            arrs["target_vel"][:] = target_vel

    _run(g, lambda a: O.step(c, arrs, a), lambda m, tasks: O.reset_tasks(c, arrs, m, 0, 0, tasks, 0), load, lambda: {k: v.copy() for k, v in arrs.items()})


def test_oracle_without_reset_pid_the_sequence_differs(golden):
    """The same replay with the input unbound (task.pid_reset = on_reset never resets through the step): the pulsed steps miss the golden."""
    g = golden(TAG)
    E, A, C, T, max_len = (int(x) for x in g["meta"])
    c = config.resolve_hns_cfg(_cfg(E, A, C, max_len))
    arrs = O.alloc_buffers(c)
    arrs["reset_pid"] = None
    t = int(np.argmax(g["root_done"][:, :, 0].any(1)))                  # first pulsed step
    st, last = _state(g, t - 1, None)
    _load_oracle(arrs, st, last)
    O.step(c, arrs, g["action"][t])
    pulsed = g["root_done"][t][:, 0]
    err = np.abs(arrs["pid_integ"][..., :3] - g["integ"][t]).max(axis=(1, 2))
    assert (err[pulsed] > 1e-3).any() and (err[~pulsed] < 1e-4).all()


@pytest.mark.gpu
def test_hip_reset_pid_across_resets(golden):
    import ctypes as C_
    import torch
    from hns_amd.env import HideAndSeek
    g = golden(TAG)
    E, A, C, T, max_len = (int(x) for x in g["meta"])
    env = HideAndSeek(_cfg(E, A, C, max_len))
    env.set_seed(0)
    env.reset()
    c = env.hcfg
    host = env.export_state()                                           # the oracle runs beside the kernel from the same loaded states

    def load(st, last, done, target_vel):
        cur = env.export_state()
        for k, v in st.items():
            cur[k] = v
        cur["pid_last_rate"][..., :3] = last
        cur["done"][:] = done
        if target_vel is not None:
            cur["target_vel"][:] = target_vel
        env.import_state(cur)
        for k in host:
            host[k][...] = env.export_state()[k]                        # incl. the derived line-of-sight column

    def step(a):
        env.step(env.rand_step_input(torch.from_numpy(np.ascontiguousarray(a)).to(env.device)))
        O.step(c, host, a)
        dev = env.export_state()
        for k in host:
            assert np.array_equal(host[k], dev[k], equal_nan=True), f"step: {k} differs from the oracle"

    def reset(mask, tasks):
        m = torch.from_numpy(np.ascontiguousarray(mask)).to(env.device)
        tk = torch.from_numpy(np.ascontiguousarray(tasks)).to(env.device)
        epoch = env.reset_epoch
        env._check(env._lib.hns_reset_tasks(env._env, C_.c_void_p(m.data_ptr()), C_.c_void_p(tk.data_ptr()), 0, C_.c_uint64(0), env._stream()), "hns_reset_tasks")
        O.reset_tasks(c, host, mask, 0, epoch, tasks, 0)
        dev = env.export_state()
        for k in host:
            assert np.array_equal(host[k], dev[k], equal_nan=True), f"reset: {k} differs from the oracle"

    _run(g, step, reset, load, env.export_state)
