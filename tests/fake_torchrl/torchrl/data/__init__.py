"""Spec classes of torchrl 0.1.1 as far as an environment and the MAPPO learner touch them: shape / dtype / device, expand, to, zero, rand,
keys(include_nested, leaves_only); a CompositeSpec carries its own `shape` (the leading, batch part)."""
import torch
from tensordict import TensorDict


def _sz(shape):
    if isinstance(shape, int):
        return torch.Size([shape])
    return torch.Size(shape)


class TensorSpec:
    def __init__(self, shape, device=None, dtype=torch.float32):
        self.shape = _sz(shape)
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.dtype = dtype

    def _args(self):
        return {}

    def expand(self, *sizes):
        sizes = sizes[0] if len(sizes) == 1 and isinstance(sizes[0], (tuple, list, torch.Size)) else sizes
        if len(sizes) < len(self.shape) or tuple(sizes[len(sizes) - len(self.shape):]) != tuple(self.shape):
            raise ValueError(f"cannot expand a spec of shape {tuple(self.shape)} to {tuple(sizes)}")
        out = self.__class__.__new__(self.__class__)
        out.__dict__.update(self.__dict__)
        out.shape = torch.Size(sizes)
        return out

    def to(self, device):
        out = self.__class__.__new__(self.__class__)
        out.__dict__.update(self.__dict__)
        out.device = torch.device(device)
        return out

    def zero(self, shape=None):
        return torch.zeros((*(shape or ()), *self.shape), dtype=self.dtype, device=self.device)

    def rand(self, shape=None):
        return torch.randn((*(shape or ()), *self.shape), device=self.device).to(self.dtype)

    def is_in(self, val):
        return tuple(val.shape[-len(self.shape):]) == tuple(self.shape) and val.dtype == self.dtype

    def __repr__(self):
        return f"{self.__class__.__name__}(shape={tuple(self.shape)}, dtype={self.dtype}, device={self.device})"


class UnboundedContinuousTensorSpec(TensorSpec):
    pass


class BoundedTensorSpec(TensorSpec):
    def __init__(self, minimum, maximum, shape=None, device=None, dtype=torch.float32):
        super().__init__(shape, device, dtype)
        self.minimum, self.maximum = minimum, maximum

    def rand(self, shape=None):
        return torch.rand((*(shape or ()), *self.shape), device=self.device) * (self.maximum - self.minimum) + self.minimum


class DiscreteTensorSpec(TensorSpec):
    def __init__(self, n, shape=None, device=None, dtype=torch.long):
        super().__init__(shape if shape is not None else (), device, dtype)
        self.n = n

    def rand(self, shape=None):
        return torch.randint(0, self.n, (*(shape or ()), *self.shape), device=self.device).to(self.dtype)


class CompositeSpec(TensorSpec):
    def __init__(self, *args, shape=None, device=None, **kwargs):
        d = dict(args[0]) if args and args[0] is not None else {}
        d.update(kwargs)
        self.shape = _sz(shape if shape is not None else ())
        self.device = torch.device(device) if device is not None else None
        self.dtype = None
        self._specs = {}
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict):
            value = CompositeSpec(value, shape=self.shape)
        if value is not None and tuple(value.shape[:len(self.shape)]) != tuple(self.shape):
            raise ValueError(f"The shape of the spec and the CompositeSpec mismatch: the first {len(self.shape)} dimensions should match but got "
                             f"spec.shape={tuple(value.shape)} and CompositeSpec.shape={tuple(self.shape)} (key {key}).")
        self._specs[key] = value

    def __getitem__(self, key):
        if isinstance(key, tuple):
            cur = self
            for k in key:
                cur = cur[k]
            return cur
        return self._specs[key]

    def __contains__(self, key):
        return key in self._specs

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def keys(self, include_nested=False, leaves_only=False):
        out = []
        for k, v in self._specs.items():
            if isinstance(v, CompositeSpec):
                if not leaves_only:
                    out.append(k)
                if include_nested:
                    out.extend((k, *(s if isinstance(s, tuple) else (s,))) for s in v.keys(True, leaves_only))
            else:
                out.append(k)
        return out

    def items(self):
        return list(self._specs.items())

    def values(self):
        return list(self._specs.values())

    def expand(self, *sizes):
        sizes = sizes[0] if len(sizes) == 1 and isinstance(sizes[0], (tuple, list, torch.Size)) else sizes
        lead = tuple(sizes)[: len(sizes) - len(self.shape)] if len(self.shape) else tuple(sizes)
        out = CompositeSpec({}, shape=tuple(sizes))
        for k, v in self._specs.items():
            out._specs[k] = v.expand(*lead, *v.shape)
        return out

    def to(self, device):
        out = CompositeSpec({}, shape=self.shape, device=device)
        for k, v in self._specs.items():
            out._specs[k] = v.to(device)
        return out

    def zero(self, shape=None):
        return TensorDict({k: v.zero(shape) for k, v in self._specs.items()}, (*(shape or ()), *self.shape))

    def rand(self, shape=None):
        return TensorDict({k: v.rand(shape) for k, v in self._specs.items()}, (*(shape or ()), *self.shape))

    def __repr__(self):
        return f"CompositeSpec(shape={tuple(self.shape)}, {self._specs})"
