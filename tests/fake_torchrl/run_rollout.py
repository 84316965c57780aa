#!/usr/bin/env python3
"""Child process of tests/test_torchrl_branch.py: the two other ways scripts/train.py consumes the env (stand-in torchrl, see README.md).

1. `evaluate()` (scripts/train.py:207-236): base_env.enable_render / .eval(), env.eval(), env.set_seed, then
   `env.rollout(max_steps, policy, callback=Every(fn, 2), auto_reset=True, break_when_any_done=False, return_contiguous=False).clone()` and the
   statistics read off `("next", "done")`.  The rollout keeps every step's tensors by reference and never resets inside the loop, so (a) the env
   must hand out NEW observation / reward / done tensors per step in eval mode, as the reference does, and (b) past the end of the episode the root
   `done` stays set and the controller is reset through it at every step (transforms.py:449-454) — the whole trajectory is replayed on the oracle.
2. Two PPO-style iterations of a small attention policy that reads `[state_self, state_others, cylinders]` in the order of the observation spec's keys
   with the pursuer's own row as the query (learning/modules/networks.py:250-298), collected through SyncDataCollector as scripts/train.py:165-205 does,
   advantages normalised with `sharding.normalise_advantages` (mappo.py:391-396 made data-parallel); the env keeps stepping bit-identically to the
   oracle while the learner holds graphs over the collected tensors."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, "oracle")]

import numpy as np
import torch
import hns_amd  # noqa: F401
from hns_amd import config, sharding, tensordict_shim
assert tensordict_shim.USING_REAL_TORCHRL and tensordict_shim.USING_REAL_TENSORDICT, "the stand-in packages were not picked up"
from hns_amd.env import HideAndSeek
from torchrl.envs import Compose, TransformedEnv
from torchrl.collectors import SyncDataCollector
import hns_oracle as O

E, A, L, T = 192, 3, 10, 6
cfg = config.make_cfg({"num_agents": A, "cylinder": {"max_num": 5, "min_num": 3}, "env": {"num_envs": E, "max_episode_length": L}},
                      algo={"use_TP_net": 0, "train_every": T})
base_env = HideAndSeek(cfg, headless=True)
env = TransformedEnv(base_env, Compose())
dev = base_env.device


class Every:                                                       # omni_drones/utils/torchrl/__init__.py: call `fn` every `steps` calls
    def __init__(self, fn, steps):
        self.fn, self.steps, self.i = fn, steps, 0

    def __call__(self, *a, **k):
        if self.i % self.steps == 0:
            self.fn(*a, **k)
        self.i += 1


class AttentionPolicy(torch.nn.Module):
    """One embedding per observation key, tokens = [self | others | cylinders] in spec-key order, one attention head with the pursuer's own token
    as the query, a mean / value head: the shape of the reference's PartialAttentionEncoder + actor, at toy size."""

    def __init__(self, spec):
        super().__init__()
        self.keys = list(spec.keys())                              # spec order IS the token order (networks.py:262-270)
        self.embed = torch.nn.ModuleDict({k: torch.nn.Linear(spec[k].shape[-1], 16) for k in self.keys})
        self.q, self.k, self.v = (torch.nn.Linear(16, 16) for _ in range(3))
        self.mean, self.value = torch.nn.Linear(16, 4), torch.nn.Linear(16, 1)
        self.log_std = torch.nn.Parameter(torch.zeros(4) - 1.0)

    def features(self, obs):
        toks = torch.cat([self.embed[k](obs[k]) for k in self.keys], dim=-2)          # [..., A, n_tokens, 16]
        q = self.q(toks[..., :1, :])                                                    # query index 0: state_self
        att = torch.softmax(q @ self.k(toks).transpose(-1, -2) / 4.0, dim=-1)
        return (att @ self.v(toks)).squeeze(-2)

    def forward(self, td, deterministic=False):
        obs = {k: td.get(("agents", "observation", k)) for k in self.keys}
        with torch.no_grad():
            f = self.features(obs)
            mean = self.mean(f)
            act = mean if deterministic else mean + torch.randn_like(mean) * self.log_std.exp()
        td.set(("agents", "action"), act)
        return td


spec = base_env.observation_spec[("agents", "observation")]
assert list(spec.keys())[0] == "state_self" and set(spec.keys()) == {"state_self", "state_others", "cylinders"}
torch.manual_seed(3)
policy = AttentionPolicy(spec).to(dev)

# ---- 1. evaluate() -------------------------------------------------------------------------------------------------------------------
frames = []
base_env.enable_render(True)
base_env.eval()
env.eval()
env.set_seed(0)
STEPS = L + 3                                                      # three steps past the end of the episode: root done set, reset_pid pulsing
with torch.no_grad():
    trajs = env.rollout(max_steps=STEPS, policy=lambda x: policy(x, deterministic=True), callback=Every(lambda *a, **k: frames.append(base_env.render(mode="rgb_array")), 2),
                        auto_reset=True, break_when_any_done=False, return_contiguous=False).clone()
done = trajs.get(("next", "done"))
assert tuple(done.shape) == (E, STEPS, 1) and done.dtype == torch.bool
first_done = torch.argmax(done.long(), dim=1).cpu()
assert (first_done == L - 1).all() and not done[:, :L - 1].any() and done[:, L - 1:].all()      # every step kept ITS done, not the last one's
rew = trajs.get(("next", "agents", "reward"))
obs_self = trajs.get(("next", "agents", "observation", "state_self"))
assert tuple(rew.shape) == (E, STEPS, A, 1) and tuple(obs_self.shape) == (E, STEPS, A, 1, 20)
assert len(frames) == (STEPS - 1 + 1) // 2                          # the callback ran (every second step_mdp)
# replay on the oracle: reset, then every action of the trajectory, never a reset in between; root done = the previous step's done
host = O.alloc_buffers(base_env.hcfg)
O.reset(base_env.hcfg, host, None, base_env.seed, 0)
acts = trajs.get(("agents", "action")).cpu().numpy()
pulsed = 0
for t in range(STEPS):
    pulsed += int(host["done"].sum())
    O.step(base_env.hcfg, host, np.ascontiguousarray(acts[:, t]))
    assert np.array_equal(host["reward"], rew[:, t, :, 0].cpu().numpy()), f"evaluate rollout step {t}: reward differs from the oracle"
    assert np.array_equal(host["obs_self"], obs_self[:, t, :, 0].cpu().numpy()), f"evaluate rollout step {t}: state_self differs from the oracle"
    assert np.array_equal(host["done"].astype(bool), done[:, t, 0].cpu().numpy())
assert pulsed == 3 * E
devs = base_env.export_state()
for k in ("drone_state", "pid_integ", "pid_last_rate", "stats", "progress"):
    assert np.array_equal(host[k], devs[k], equal_nan=True), f"{k} differs from the oracle after the evaluate rollout"
# scripts/train.py:236-254 in meaning: re-enable / reset, first_done, take_first_episode over ("next", "stats"), the video array
base_env.enable_render(not True)                                    # (cfg.headless = True)
env.reset()


def take_first_episode(tensor):
    indices = first_done.reshape(first_done.shape + (1,) * (tensor.ndim - 2))
    return torch.take_along_dim(tensor, indices, dim=1).reshape(-1)


traj_stats = {k: take_first_episode(v) for k, v in trajs[("next", "stats")].cpu().items()}
info = {"eval/stats." + k: torch.nanmean(v.float()).item() for k, v in traj_stats.items()}
assert len(info) == 24 and all(v.shape == (E,) for v in traj_stats.values())
assert np.isfinite(list(info.values())).all()
assert all(isinstance(f, np.ndarray) and f.dtype == np.uint8 and f.ndim == 3 and f.shape[2] == 3 for f in frames)
video_array = np.stack(frames).transpose(0, 3, 1, 2)                # [N, 3, H, W]: what wandb.Video(video_array, fps=..., format="mp4") takes
assert video_array.shape == (len(frames), 3) + frames[0].shape[:2] and video_array.dtype == np.uint8
assert len({f.tobytes() for f in frames}) == len(frames)            # the drones moved between frames: no two frames are the same picture
n_frames = len(frames)
frames.clear()

# ---- 2. two PPO-style iterations through the collector -----------------------------------------------------------------------------------
base_env.train()
env.train()
env.reset()
opt = torch.optim.Adam(policy.parameters(), lr=1e-3)
frames_per_batch = E * T
collector = SyncDataCollector(env, policy=policy, frames_per_batch=frames_per_batch, total_frames=frames_per_batch * 2, device=cfg.sim.device, return_same_td=True)
O.reset(base_env.hcfg, host, None, base_env.seed, base_env.reset_epoch - 1)
epoch = base_env.reset_epoch
losses = []
for it, data in enumerate(collector):
    obs = {k: data.get(("agents", "observation", k)) for k in policy.keys}           # [E, T, A, ...]
    act, r = data.get(("agents", "action")), data.get(("next", "agents", "reward"))
    f = policy.features(obs)
    value = policy.value(f)
    adv, rate = sharding.normalise_advantages((r - value).detach(), base_env.stats["success"])
    assert abs(float(adv.mean())) < 1e-4 and abs(float(adv.std()) - 1.0) < 1e-3 and 0.0 <= rate <= 1.0
    logp = -(((act - policy.mean(f)) / policy.log_std.exp()) ** 2).sum(-1, keepdim=True) * 0.5 - policy.log_std.sum()
    loss = -(logp * adv).mean() + 0.5 * ((r - value) ** 2).mean()
    opt.zero_grad()
    loss.backward()                                                                 # graphs over the collected tensors: the env's in-place buffers are not in them
    opt.step()
    losses.append(float(loss))
    a_np = act.detach().cpu().numpy()
    for t in range(T):
        O.step(base_env.hcfg, host, np.ascontiguousarray(a_np[:, t]))
        assert np.array_equal(host["reward"], r[:, t, :, 0].detach().cpu().numpy()), f"iteration {it} step {t}: reward differs from the oracle"
        if host["done"].any():
            O.reset(base_env.hcfg, host, host["done"].copy(), base_env.seed, epoch)
            epoch += 1
assert all(np.isfinite(losses)) and len(losses) == 2
print(json.dumps({"evaluate_steps": STEPS, "reset_pid_pulses": pulsed, "frames": n_frames, "video_shape": list(video_array.shape),
                  "eval_stats": len(info), "ppo_iterations": len(losses)}))
