"""EnvBase / TransformedEnv / Compose of torchrl 0.1.1, reduced to the contract they enforce on an environment (see ../../README.md)."""
import torch
from torch import nn
from tensordict import TensorDict, TensorDictBase
from torchrl.data import CompositeSpec, DiscreteTensorSpec, TensorSpec
from .utils import step_mdp  # noqa: F401


class EnvBase(nn.Module):
    def __init__(self, device="cpu", dtype=None, batch_size=None, run_type_checks=False):
        super().__init__()
        self.__dict__["_observation_spec"] = None
        self.__dict__["_input_spec"] = None
        self.__dict__["_reward_spec"] = None
        self.__dict__["_done_spec"] = None
        self.__dict__["device"] = torch.device(device)
        self.__dict__["batch_size"] = torch.Size(batch_size if batch_size is not None else [])
        self.run_type_checks = run_type_checks
        self.is_closed = False

    # ---- specs: shape[: len(batch_size)] must equal batch_size -------------------------------------------------------
    def _lead_ok(self, value, what):
        nb = len(self.batch_size)
        if tuple(value.shape[:nb]) != tuple(self.batch_size):
            raise ValueError(f"The value of spec.shape ({tuple(value.shape)}) must match the env batch size ({tuple(self.batch_size)}) [{what}].")

    @property
    def observation_spec(self):
        return self.__dict__["_observation_spec"]

    @observation_spec.setter
    def observation_spec(self, value):
        if not isinstance(value, CompositeSpec):
            raise TypeError("The type of an observation_spec must be Composite.")
        self._lead_ok(value, "observation_spec")
        self.__dict__["_observation_spec"] = value.to(self.device)

    @property
    def action_spec(self):
        return self.input_spec["_action_spec"]

    @action_spec.setter
    def action_spec(self, value):
        if not isinstance(value, TensorSpec):
            raise TypeError("action_spec must be a TensorSpec")
        self._lead_ok(value, "action_spec")
        self.__dict__["_input_spec"] = CompositeSpec({"_action_spec": value.to(self.device)}, shape=self.batch_size)

    @property
    def input_spec(self):
        return self.__dict__["_input_spec"]

    @input_spec.setter
    def input_spec(self, value):
        raise RuntimeError("input_spec is protected: set action_spec")

    @property
    def reward_spec(self):
        return self.__dict__["_reward_spec"]

    @reward_spec.setter
    def reward_spec(self, value):
        if not isinstance(value, TensorSpec):
            raise TypeError("reward_spec must be a TensorSpec")
        self._lead_ok(value, "reward_spec")
        self.__dict__["_reward_spec"] = value.to(self.device)

    @property
    def done_spec(self):
        if self.__dict__["_done_spec"] is None:
            self.__dict__["_done_spec"] = DiscreteTensorSpec(2, (*self.batch_size, 1), device=self.device, dtype=torch.bool)
        return self.__dict__["_done_spec"]

    @done_spec.setter
    def done_spec(self, value):
        self._lead_ok(value, "done_spec")
        self.__dict__["_done_spec"] = value.to(self.device)

    @property
    def reward_key(self):
        r = self.reward_spec
        if isinstance(r, CompositeSpec):
            keys = r.keys(True, True)
            if len(keys) != 1:
                raise RuntimeError("one reward key expected")
            return keys[0]
        return "reward"

    # ---- the public calls --------------------------------------------------------------------------------------------
    def _assert_tensordict_shape(self, tensordict):
        if tuple(tensordict.batch_size) != tuple(self.batch_size):
            raise RuntimeError(f"Expected a tensordict with shape==env.shape, got {tuple(tensordict.batch_size)} and {tuple(self.batch_size)}")

    def step(self, tensordict):
        self._assert_tensordict_shape(tensordict)
        tensordict.lock_()                                     # _step must not set new keys on its input
        try:
            out = self._step(tensordict)
        finally:
            tensordict.unlock_()
        if out is tensordict:
            raise RuntimeError("EnvBase._step should return outplace changes to the input tensordict.")
        if not isinstance(out, TensorDictBase):
            raise TypeError("_step must return a tensordict")
        nxt = out.get("next")                                  # KeyError if missing
        self._assert_tensordict_shape(out)
        rew = nxt.get(self.reward_key)
        done = nxt.get("done")
        if done.dtype != torch.bool or tuple(done.shape) != (*self.batch_size, 1):
            raise RuntimeError(f"done must be a bool tensor of shape {(*self.batch_size, 1)}, got {done.dtype} {tuple(done.shape)}")
        if tuple(rew.shape[: len(self.batch_size)]) != tuple(self.batch_size):
            raise RuntimeError("reward does not carry the batch size")
        if self.run_type_checks:
            for k in self.observation_spec.keys(True, True):
                if not self.observation_spec[k].is_in(nxt.get(k)):
                    raise TypeError(f"observation {k} does not match its spec")
        tensordict.update(out)
        return tensordict

    def reset(self, tensordict=None, **kwargs):
        _reset = None
        if tensordict is not None:
            self._assert_tensordict_shape(tensordict)
            _reset = tensordict.get("_reset", None)
            if _reset is not None and _reset.dtype != torch.bool:
                raise TypeError("_reset must be a bool mask")
        td_reset = self._reset(tensordict, **kwargs)
        if td_reset is tensordict and tensordict is not None:
            raise RuntimeError("EnvBase._reset should return outplace changes to the input tensordict.")
        if not isinstance(td_reset, TensorDictBase):
            raise RuntimeError(f"env._reset returned an object of type {type(td_reset)} but a TensorDict was expected.")
        self._assert_tensordict_shape(td_reset)
        if "done" not in td_reset.keys():
            td_reset.set("done", torch.zeros((*self.batch_size, 1), dtype=torch.bool, device=self.device))
        done = td_reset.get("done")
        if (_reset is None and done.any()) or (_reset is not None and done[_reset.reshape(done.shape)].any()):
            raise RuntimeError(f"Env {self} was done after reset on specified '_reset' dimensions. This is (currently) not allowed.")
        if tensordict is not None:
            tensordict.update(td_reset)
        else:
            tensordict = td_reset
        return tensordict

    def rollout(self, max_steps, policy=None, callback=None, auto_reset=True, auto_cast_to_device=False, break_when_any_done=True,
                return_contiguous=True, tensordict=None):
        """torchrl 0.1.1 `EnvBase.rollout`: reset (auto_reset), then policy -> step -> keep `tensordict.clone(False)` (the step's tensors BY
        REFERENCE) -> step_mdp -> callback; no reset inside the loop — with break_when_any_done=False finished envs keep being stepped with
        their root `done` set; the kept steps are stacked along a new last batch dimension, lazily with return_contiguous=False."""
        from .utils import step_mdp
        if auto_reset:
            if tensordict is not None:
                raise RuntimeError("tensordict cannot be provided when auto_reset is True")
            tensordict = self.reset()
        elif tensordict is None:
            raise RuntimeError("tensordict must be provided when auto_reset is False")
        if policy is None:
            def policy(td):
                return td.update(self.action_spec.rand())
        kept = []
        for i in range(max_steps):
            tensordict = policy(tensordict)
            tensordict = self.step(tensordict)
            kept.append(tensordict.clone(False))
            if (break_when_any_done and bool(tensordict.get(("next", "done")).any())) or i == max_steps - 1:
                break
            tensordict = step_mdp(tensordict, keep_other=True, exclude_action=False, exclude_reward=True)
            if callback is not None:
                callback(self, tensordict)
        out = LazyStackedTensorDict(kept, len(self.batch_size))
        return out.contiguous() if return_contiguous else out

    def set_seed(self, seed=None, static_seed=False):
        if seed is not None:
            torch.manual_seed(seed)
        self._set_seed(seed)
        return seed

    def rand_step(self, tensordict=None):
        if tensordict is None:
            tensordict = TensorDict({}, self.batch_size, self.device)
        tensordict.update(self.action_spec.rand() if isinstance(self.action_spec, CompositeSpec) else {"action": self.action_spec.rand()})
        return self.step(tensordict)

    def close(self):
        self.is_closed = True

    def __repr__(self):
        return f"{self.__class__.__name__}(batch_size={tuple(self.batch_size)}, device={self.device})"


class LazyStackedTensorDict:
    """What `torch.stack(list_of_tensordicts, dim)` gives in tensordict 0.1.2: the tensordicts are kept; an entry is stacked when it is read."""

    def __init__(self, tds, dim):
        self.tensordicts, self.stack_dim = list(tds), dim
        b = tuple(tds[0].batch_size)
        self.batch_size = torch.Size((*b[:dim], len(tds), *b[dim:]))

    def get(self, key, default=None):
        vals = [td.get(key, None) for td in self.tensordicts]
        if any(v is None for v in vals):
            if default is None:
                raise KeyError(key)
            return default
        if isinstance(vals[0], TensorDictBase):
            return LazyStackedTensorDict(vals, self.stack_dim)
        return torch.stack(vals, self.stack_dim)

    def __getitem__(self, key):
        if isinstance(key, (str, tuple)) and (isinstance(key, str) or all(isinstance(k, str) for k in key)):
            return self.get(key)
        raise NotImplementedError("only key access")

    def keys(self, *a, **k):
        return self.tensordicts[0].keys(*a, **k)

    def items(self, *a, **k):
        return [(key, self.get(key)) for key in self.keys(*a, **k)]

    def to(self, device):
        return LazyStackedTensorDict([td.to(device) for td in self.tensordicts], self.stack_dim)

    def cpu(self):
        return self.to("cpu")

    def clone(self, recurse=True):
        return LazyStackedTensorDict([td.clone(recurse) for td in self.tensordicts], self.stack_dim)

    def contiguous(self):
        out = TensorDict({}, self.batch_size, self.tensordicts[0].device)
        for k in self.tensordicts[0].keys(True, True):
            out.set(k, self.get(k))
        return out


class Transform(nn.Module):
    def inv(self, tensordict):
        return self._inv_call(tensordict)

    def _inv_call(self, tensordict):
        return tensordict

    def _call(self, tensordict):
        return tensordict

    def transform_input_spec(self, spec):
        return spec

    def transform_observation_spec(self, spec):
        return spec

    def transform_reward_spec(self, spec):
        return spec


class Compose(Transform):
    def __init__(self, *transforms):
        super().__init__()
        self.transforms = nn.ModuleList(transforms)

    def _inv_call(self, tensordict):
        for t in reversed(self.transforms):
            tensordict = t._inv_call(tensordict)
        return tensordict

    def _call(self, tensordict):
        for t in self.transforms:
            tensordict = t._call(tensordict)
        return tensordict


class TransformedEnv(EnvBase):
    def __init__(self, env, transform=None):
        super().__init__(device=env.device, batch_size=env.batch_size)
        self.__dict__["base_env"] = env
        self.transform = transform if transform is not None else Compose()

    # the specs are the base env's, seen through the transform
    @property
    def observation_spec(self):
        return self.transform.transform_observation_spec(self.base_env.observation_spec)

    @property
    def action_spec(self):
        return self.transform.transform_input_spec(self.base_env.input_spec)["_action_spec"]

    @property
    def input_spec(self):
        return self.transform.transform_input_spec(self.base_env.input_spec)

    @property
    def reward_spec(self):
        return self.transform.transform_reward_spec(self.base_env.reward_spec)

    @property
    def done_spec(self):
        return self.base_env.done_spec

    def _step(self, tensordict):
        tensordict = tensordict.clone(False)                   # shallow: the transform's inverse may add keys
        tensordict_in = self.transform.inv(tensordict)
        out = self.base_env._step(tensordict_in)
        out.set("next", self.transform._call(out.get("next")))
        return out

    def _reset(self, tensordict=None, **kwargs):
        out = self.base_env._reset(tensordict, **kwargs)
        return self.transform._call(out)

    def _set_seed(self, seed):
        return self.base_env._set_seed(seed)

    def close(self):
        self.base_env.close()
        self.is_closed = True

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            base = self.__dict__.get("base_env")
            if base is None or name.startswith("__"):
                raise
            return getattr(base, name)
