#!/usr/bin/env python3
"""Child process of tests/test_torchrl_branch.py: HideAndSeek as a (stand-in) torchrl EnvBase under TransformedEnv + SyncDataCollector,
wired as scripts/train.py:165-205 wires the reference's env; the collected rollouts are replayed on the CPU oracle.

argv: [use_TP_net 0|1] [motor] — `motor`: the reference's task file unchanged (`action_transform: PIDrate`, cfg/task/HideAndSeek.yaml:16): train.py:165-171
puts the controller transform in front; here a stand-in that does what `PIDRateController._inv_call` does (utils/torchrl/transforms.py:425-459) with the
oracle's restatement of the controller.  The env derives `action_input: motor` from the same key; the oracle follows with its FUSED controller from the raw
policy actions, so the replay also proves motor == policy."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, "oracle")]

import numpy as np
import torch
import hns_amd  # noqa: F401
from hns_amd import config, tensordict_shim
assert tensordict_shim.USING_REAL_TORCHRL and tensordict_shim.USING_REAL_TENSORDICT, "the stand-in packages were not picked up"
from hns_amd.env import HideAndSeek
from tensordict import TensorDict
from torchrl.envs import Compose, EnvBase, Transform, TransformedEnv
from torchrl.collectors import SyncDataCollector
import hns_oracle as O

use_tp = int(sys.argv[1]) if len(sys.argv) > 1 else 0
motor = len(sys.argv) > 2 and sys.argv[2] == "motor"
E, A, L, T = 256, 3, 12, 8
if motor:
    import tempfile
    # a task file as the reference ships it — `action_transform: PIDrate`, no key of this build — read the way hydra's composed config arrives
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        f.write(f"name: HideAndSeek\naction_transform: PIDrate\nnum_agents: {A}\ncylinder:\n  max_num: 5\n  min_num: 3\n"
                f"env:\n  num_envs: {E}\n  max_episode_length: {L}\n")
    cfg = config.load_cfg(f.name)
    os.unlink(f.name)
    cfg.algo.use_TP_net, cfg.algo.train_every = use_tp, T
    assert "action_input" not in cfg.task
else:
    cfg = config.make_cfg({"num_agents": A, "cylinder": {"max_num": 5, "min_num": 3}, "env": {"num_envs": E, "max_episode_length": L}},
                          algo={"use_TP_net": use_tp, "train_every": T})
from omni_drones.envs.isaac_env import IsaacEnv                   # scripts/train.py:100,110-111: the class comes from the registry, by the task's name
base_env = IsaacEnv.REGISTRY[cfg.task.name](cfg, headless=True)
assert isinstance(base_env, HideAndSeek) and base_env.action_input == ("motor" if motor else "policy")
assert isinstance(base_env, EnvBase) and tuple(base_env.batch_size) == (E,)
assert tuple(base_env.observation_spec.shape) == (E,) and tuple(base_env.input_spec["_action_spec"].shape) == (E,)
assert tuple(base_env.observation_spec[("agents", "observation", "state_self")].shape) == (E, A, 1, 35 if use_tp else 20)


class StandInPIDRate(Transform):
    """What scripts/train.py:165-171 puts in front with `action_transform: PIDrate`: reads ("info","drone_state"), the action, ("info","prev_action") and
    the root `done` of the stepped tensordict; leaves rotor commands in ("agents","action") and the keys hideandseek.py:729-731 reads back
    (transforms.py:425-459).  The arithmetic is the oracle's restatement of `_inv_call` + PIDRateController.forward (pinned by g_pid), on the host."""

    def __init__(self, hcfg, device):
        super().__init__()
        self.c, self.dev = hcfg, device
        self.integ = np.zeros((E * A, 3), np.float32)             # the controller's own state (lee_position_controller.py:468-470)
        self.last = np.zeros((E * A, 3), np.float32)

    def _inv_call(self, td):
        ds = td[("info", "drone_state")][..., :13].cpu().numpy().reshape(E * A, 13)
        act = td[("agents", "action")].cpu().numpy()
        prev = td[("info", "prev_action")].cpu().numpy()
        reset_pid = np.repeat(td["done"].reshape(E).cpu().numpy().astype(np.uint8), A)          # .expand(-1, num_drones), transforms.py:453
        r = O.ctbr_pid(self.c, act, ds[:, 3:7], ds[:, 10:13], reset_pid, prev, self.integ, self.last)
        self.integ, self.last = r["integ"], r["last"]
        td.set(("stats", "action_error_order1"), torch.from_numpy(r["aerr"].reshape(E, A)).to(self.dev))
        td.set(("info", "prev_action"), torch.from_numpy(r["prev_action"].reshape(E, A, 4)).to(self.dev))
        td.set(("agents", "action"), torch.from_numpy(r["cmd"].reshape(E, A, 4)).to(self.dev))
        td.set("ctbr", torch.from_numpy(r["ctbr"].reshape(E, A, 4)).to(self.dev))
        td.set("target_rate", torch.from_numpy(r["target_rate"].reshape(E, A, 3)).to(self.dev))
        return td


transforms = [StandInPIDRate(base_env.hcfg, base_env.device)] if motor else []       # action_transform: none (cfg/task/HideAndSeek_hip.yaml) -> nothing
env = TransformedEnv(base_env, Compose(*transforms)).train()
env.set_seed(0)
agent_spec = env.agent_spec["drone"]                          # reached through the wrapper, as train.py:176 does
assert agent_spec.n == A

gen = torch.Generator(device=base_env.device).manual_seed(5)


def policy(td):
    td.set(("agents", "action"), torch.randn(E, A, 4, generator=gen, device=base_env.device))
    return td


frames_per_batch = env.num_envs * int(cfg.algo.train_every)
collector = SyncDataCollector(env, policy=policy, frames_per_batch=frames_per_batch, total_frames=frames_per_batch * 4,
                              device=cfg.sim.device, return_same_td=True)
# the oracle follows: full reset, then every collected action; masked resets where the collector issued them
ocfg = base_env.hcfg.copy()
ocfg.action_input = 0                                             # the oracle runs the FUSED controller on the raw policy actions, whichever input the env takes
host = O.alloc_buffers(ocfg)
O.reset(ocfg, host, None, base_env.seed, 0)
epoch, n_resets, first = 1, 0, None
# scripts/train.py:113-116,193-196: the statistics `EpisodeStats` follows are the leaves of the observation spec under "stats"
stats_keys = [k for k in base_env.observation_spec.keys(True, True) if isinstance(k, tuple) and k[0] == "stats"]
assert len(stats_keys) == 24 and ("stats", "success") in stats_keys
from hns_amd import abi
episodes_seen, pending = 0, None                                  # pending: (env mask, oracle statistics) of an episode that ended on a rollout's last step
for i, data in enumerate(collector):
    assert tuple(data.batch_size) == (E, T)
    if first is None:
        first = data
    assert data is first                                          # return_same_td
    rew, done = data.get(("next", "agents", "reward")), data.get(("next", "done"))
    assert tuple(rew.shape) == (E, T, A, 1) and tuple(done.shape) == (E, T, 1) and done.dtype == torch.bool
    assert tuple(data.get(("next", "agents", "observation", "state_self")).shape) == (E, T, A, 1, 35 if use_tp else 20)
    if not motor:                                                 # (motor: the transform sets them on TransformedEnv._step's shallow clone, as in the reference)
        assert tuple(data.get(("stats", "action_error_order1")).shape) == (E, T, A)
    assert tuple(data.get(("info", "prev_action")).shape) == (E, T, A, 4)
    assert "_reset" not in data.keys()
    acts = data.get(("agents", "action")).cpu().numpy()
    # EpisodeStats.__call__ (scripts/train.py:58-72) in meaning: the ROOT statistics one step behind a `done` are the finished episode's — the reset that
    # followed handed the pre-reset statistics back (isaac_env.py:216,223-224) and step_mdp carried them into the next frame:
    #     tensordict.select(*in_keys)[:, 1:][done_or_truncated[:, :-1]]
    truncated = data.get(("next", "truncated"), None)
    assert truncated is None                                      # (the reference's step does not write it either: `done.clone()` is what EpisodeStats uses)
    dmask = done.squeeze(-1)                                      # [E, T]
    picked = {k: data.get(k)[:, 1:][dmask[:, :-1]].clone() for k in stats_keys}
    episodes_seen += int(dmask.sum())
    expect = {name: [] for name in abi.STAT_NAMES}                # the oracle's statistics of the envs that finished, in (env-major, time) order of the mask
    ends = []
    for t in range(T):
        O.step(ocfg, host, np.ascontiguousarray(acts[:, t]))
        assert np.array_equal(host["reward"], rew[:, t, :, 0].cpu().numpy()), f"rollout {i} step {t}: reward differs from the oracle"
        assert np.array_equal(host["done"].astype(bool), done[:, t, 0].cpu().numpy())
        if host["done"].any():
            if t < T - 1:
                ends.append((t, host["done"].astype(bool).copy(), host["stats"].copy()))
            O.reset(ocfg, host, host["done"].copy(), base_env.seed, epoch)
            epoch += 1
            n_resets += 1
    # boolean-mask indexing of [E, T-1] walks env-major: rebuild the oracle's picks in that order
    for j, name in enumerate(abi.STAT_NAMES):
        want = []
        for e in range(E):
            for t, dn, st in ends:
                if dn[e]:
                    want.append(st[j, e])
        got = picked[("stats", name)].reshape(-1).cpu().numpy()
        assert got.shape[0] == len(want), (name, got.shape, len(want))
        assert np.array_equal(got, np.asarray(want, dtype=np.float32), equal_nan=True), f"rollout {i}: finished-episode statistic {name} differs from the oracle's"
    if ends:
        # `pop()` (train.py:74-77) averages them; e.g. no finished episode reports a first capture beyond its length
        assert (picked[("stats", "first_capture_step")] <= L).all() and picked[("stats", "success")].shape[0] == sum(int(dn.sum()) for _, dn, _ in ends)
    if use_tp:
        # learning/mappo.py:407-427 + update_TP (:252-268) in meaning: the learner trains the env's predictor on windows of the collected ground truth.
        #   windows = TP_groundtruth.unfold(1, F + 1, window_step).transpose(2, 3)[:, :, 1:]; mask = TP_done[:, :n] ...; masked_select(...).view(batch, -1, F, 3)
        F, ws = base_env.TP.future_predcition_step, base_env.TP.window_step
        gt = data.get(("next", "agents", "TP", "TP_groundtruth"))                 # [E, T, 3] (hideandseek.py:840-842: scaled to (-1, 1))
        tin = data.get(("next", "agents", "TP", "TP_input"))                      # [E, T, 10, 16]
        tdone = data.get(("next", "agents", "TP", "TP_done"))                     # [E, T, 1] bool
        assert tuple(gt.shape) == (E, T, 3) and tuple(tin.shape) == (E, T, 10, 16) and tuple(tdone.shape) == (E, T, 1) and tdone.dtype == torch.bool
        windows = gt.unfold(dimension=1, size=F + 1, step=ws).transpose(2, 3)[:, :, 1:]
        batch, _, future_step, pos_dim = windows.shape
        assert (future_step, pos_dim) == (F, 3)
        mask = tdone[:, :windows.shape[1]].squeeze(-1).unsqueeze(-1).unsqueeze(-1).expand_as(windows).bool()
        selected = torch.masked_select(windows, mask).view(batch, -1, future_step, pos_dim)     # needs the same count in every env: episodes are lock step
        n_sel = selected.shape[1]
        if n_sel:
            x, y = tin[:, :n_sel].reshape(-1, 10, 16), selected.reshape(-1, F * 3)
            assert bool(torch.isfinite(y).all())     # (xy / (0.5 arena_size): up to +-2 — the reference's "clip to (-1, 1)" comment, :839, is not followed by a clip)
            opt = torch.optim.Adam(base_env.TP.parameters(), lr=1e-3)
            loss = torch.nn.functional.mse_loss(base_env.TP(x.clone()), y.clone())
            opt.zero_grad()
            loss.backward()
            opt.step()                                                            # in-place update: the env re-packs the operand image at its next step
            assert np.isfinite(float(loss))
            tp_updates = globals().get("tp_updates", 0) + 1
dev = base_env.export_state()
for k in ("drone_state", "target_pos", "progress", "stats"):
    assert np.array_equal(host[k], dev[k], equal_nan=True), f"{k} differs from the oracle after the rollouts"
# reset(td) with a tensordict that carries no `_reset` (tensordict 0.1.x: get() raises on a missing key)
td = env.reset(TensorDict({}, [E], device=base_env.device))
assert not td.get("done").any()
print(json.dumps({"action_input": base_env.action_input, "rollouts": i + 1, "frames": (i + 1) * E * T, "masked_resets": n_resets, "use_tp": use_tp, "episodes_seen": episodes_seen, "stats_keys": len(stats_keys),
                  "tp_updates": globals().get("tp_updates", 0)}))
