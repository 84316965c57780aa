"""SyncDataCollector of torchrl 0.1.1 as scripts/train.py:197-205 uses it: frames_per_batch = num_envs * train_every, return_same_td=True."""
import torch
from tensordict import TensorDict
from torchrl.envs.utils import step_mdp


def _stack_into(buf, t, td):
    for k in td.keys(True, True):
        v = td.get(k)
        dst = buf.get(k, None)
        if dst is None:
            dst = torch.empty((v.shape[0], buf.batch_size[1], *v.shape[1:]), dtype=v.dtype, device=v.device)
            buf.set(k, dst)
        dst[:, t].copy_(v)


class SyncDataCollector:
    def __init__(self, create_env_fn, policy, frames_per_batch, total_frames=-1, device=None, return_same_td=False, split_trajs=False,
                 postproc=None, exploration_mode=None, reset_at_each_iter=False):
        self.env = create_env_fn
        self.policy = policy
        self.frames_per_batch = frames_per_batch
        self.total_frames = total_frames
        self.return_same_td = return_same_td
        self.split_trajs, self.postproc = split_trajs, postproc
        self._exclude_private_keys = True
        self.n_env = self.env.batch_size.numel()
        if frames_per_batch % self.n_env:
            raise ValueError("frames_per_batch must be a multiple of the number of envs")
        self._steps = frames_per_batch // self.n_env
        self._tensordict = self.env.reset()
        self._tensordict_out = None
        self._frames = 0

    @torch.no_grad()
    def rollout(self):
        if self._tensordict_out is None or not self.return_same_td:
            self._tensordict_out = TensorDict({}, [self.n_env, self._steps], self.env.device)
        for t in range(self._steps):
            self._tensordict = self.policy(self._tensordict)
            self._tensordict = self.env.step(self._tensordict)
            _stack_into(self._tensordict_out, t, self._tensordict)
            done = self._tensordict.get(("next", "done"))
            self._tensordict = step_mdp(self._tensordict)
            if done.any():
                self._tensordict.set("_reset", done)
                self.env.reset(self._tensordict)
                self._tensordict.exclude("_reset", inplace=True)
        return self._tensordict_out

    def iterator(self):
        while True:
            out = self.rollout()
            self._frames += out.numel()
            if self._exclude_private_keys:
                out = out.exclude(*[k for k in out.keys() if isinstance(k, str) and k.startswith("_")], inplace=True)
            yield out if self.return_same_td else out.clone()
            if 0 < self.total_frames <= self._frames:
                break

    def __iter__(self):
        return self.iterator()

    def shutdown(self):
        self.env.close()
