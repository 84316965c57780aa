"""Minimal TensorDict / spec stand-ins used when `tensordict` / `torchrl` are not installed.

Only the surface the reference's env and collector touch is provided (nested keys as tuples,
`get/set/update/clone/select/exclude/keys/items`, `batch_size`, leaf indexing) — reference usage:
omni_drones/envs/isaac_env.py:141-151,210-240, hideandseek.py:726-731,894-917,1060-1065.
If the real packages are importable they are used instead (see `TensorDict` below).
"""
import torch

try:  # pragma: no cover - not installed in the build image
    from tensordict import TensorDict as _RealTensorDict  # type: ignore
except Exception:  # noqa: BLE001
    _RealTensorDict = None


class _ShimTensorDict(dict):
    def __init__(self, source=None, batch_size=None, device=None):
        super().__init__()
        self.batch_size = torch.Size(batch_size if batch_size is not None else [])
        self.device = torch.device(device) if device is not None else None
        for k, v in (source or {}).items():
            self.set(k, v)

    # -- nested access ---------------------------------------------------------------------
    def _wrap(self, v):
        if isinstance(v, dict) and not isinstance(v, _ShimTensorDict):
            return _ShimTensorDict(v, self.batch_size, self.device)
        return v

    def __getitem__(self, key):
        if isinstance(key, tuple) and key and all(isinstance(k, str) for k in key):
            cur = self
            for k in key:
                cur = cur[k]                 # through __getitem__: nested lazy entries (env._LazyState) resolve
            return cur
        if isinstance(key, str):
            return dict.__getitem__(self, key)
        return self._index(key)

    def __setitem__(self, key, value):
        if isinstance(key, (str, tuple)) and (isinstance(key, str) or all(isinstance(k, str) for k in key)):
            self.set(key, value)
        else:  # td[mask_or_ids] = scalar / td
            for k, v in dict.items(self):
                if isinstance(v, _ShimTensorDict):
                    v[key] = value[k] if isinstance(value, dict) else value
                else:
                    v[key] = value[k] if isinstance(value, dict) else value

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def set(self, key, value, inplace=False):
        value = self._wrap(value)
        if isinstance(key, tuple):
            cur = self
            for k in key[:-1]:
                if k not in cur:
                    dict.__setitem__(cur, k, _ShimTensorDict({}, self.batch_size, self.device))
                cur = dict.__getitem__(cur, k)
            dict.__setitem__(cur, key[-1], value)
        else:
            dict.__setitem__(self, key, value)
        return self

    def update(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), _ShimTensorDict):
                self[k].update(v)
            else:
                self.set(k, v)
        return self

    def keys(self, include_nested=False, leaves_only=False):
        if not include_nested:
            return dict.keys(self)
        out = []
        for k, v in dict.items(self):
            if isinstance(v, _ShimTensorDict):
                if not leaves_only:
                    out.append(k)
                out.extend((k, *(sub if isinstance(sub, tuple) else (sub,))) for sub in v.keys(True, leaves_only))
            else:
                out.append(k)
        return out

    # -- whole-tree ops ----------------------------------------------------------------------
    def _map(self, fn, batch_size=None):
        out = _ShimTensorDict({}, self.batch_size if batch_size is None else batch_size, self.device)
        for k, v in dict.items(self):
            dict.__setitem__(out, k, v._map(fn, batch_size) if isinstance(v, _ShimTensorDict) else fn(v))
        return out

    def clone(self, recurse=True):
        return self._map(lambda t: t.clone() if recurse else t)

    def to(self, device):
        out = self._map(lambda t: t.to(device))
        out.device = torch.device(device)
        return out

    def cpu(self):
        return self.to("cpu")

    def _index(self, idx):
        probe = torch.empty(self.batch_size, device="meta")[idx]
        return self._map(lambda t: t[idx], probe.shape)

    def select(self, *keys):
        out = _ShimTensorDict({}, self.batch_size, self.device)
        for k in keys:
            out.set(k, self[k])
        return out

    def exclude(self, *keys):
        out = self._map(lambda t: t)
        for k in keys:
            if isinstance(k, tuple):
                cur = out
                for kk in k[:-1]:
                    cur = dict.__getitem__(cur, kk)
                dict.pop(cur, k[-1], None)
            else:
                dict.pop(out, k, None)
        return out

    @property
    def shape(self):
        return self.batch_size

    def __repr__(self):
        def shapes(d):
            return {k: (shapes(v) if isinstance(v, _ShimTensorDict) else tuple(v.shape)) for k, v in dict.items(d)}
        return f"TensorDict(batch_size={tuple(self.batch_size)}, fields={shapes(self)})"


TensorDict = _RealTensorDict if _RealTensorDict is not None else _ShimTensorDict
USING_REAL_TENSORDICT = _RealTensorDict is not None


# ---- specs: shape/dtype/bounds records with the attributes the learner reads -------------------
class TensorSpec:
    def __init__(self, shape, dtype=torch.float32, device=None, low=None, high=None):
        self.shape = torch.Size(shape)
        self.dtype = dtype
        self.device = device
        self.low, self.high = low, high

    def expand(self, *sizes):
        sizes = sizes[0] if len(sizes) == 1 and isinstance(sizes[0], (tuple, list, torch.Size)) else sizes
        return TensorSpec((*sizes, *self.shape), self.dtype, self.device, self.low, self.high)

    def to(self, device):
        return TensorSpec(self.shape, self.dtype, device, self.low, self.high)

    def zero(self):
        return torch.zeros(self.shape, dtype=self.dtype, device=self.device)

    def rand(self):
        if self.low is not None:
            return torch.rand(self.shape, device=self.device) * (self.high - self.low) + self.low
        return torch.randn(self.shape, device=self.device)

    def __repr__(self):
        return f"TensorSpec(shape={tuple(self.shape)}, dtype={self.dtype})"


class CompositeSpec(dict):
    def __init__(self, d=None, shape=()):
        super().__init__()
        self.shape = torch.Size(shape)
        for k, v in (d or {}).items():
            self[k] = CompositeSpec(v) if isinstance(v, dict) and not isinstance(v, CompositeSpec) else v

    def __getitem__(self, key):
        if isinstance(key, tuple):
            cur = self
            for k in key:
                cur = cur[k]                 # through __getitem__: nested lazy entries (env._LazyState) resolve
            return cur
        return dict.__getitem__(self, key)

    def expand(self, *sizes):
        sizes = sizes[0] if len(sizes) == 1 and isinstance(sizes[0], (tuple, list, torch.Size)) else sizes
        out = CompositeSpec({}, (*sizes, *self.shape))
        for k, v in self.items():
            dict.__setitem__(out, k, v.expand(*sizes))
        return out

    def to(self, device):
        out = CompositeSpec({}, self.shape)
        for k, v in self.items():
            dict.__setitem__(out, k, v.to(device))
        return out

    def zero(self):
        return _ShimTensorDict({k: v.zero() for k, v in self.items()}, self.shape)

    def rand(self):
        return _ShimTensorDict({k: v.rand() for k, v in self.items()}, self.shape)


# ---- one construction surface for both worlds ---------------------------------------------------------------------
# With torchrl importable the env is a real `torchrl.envs.EnvBase` and its specs are real torchrl specs (the caller —
# scripts/train.py:165-205, utils/torchrl/collector.py, learning/mappo.py — type-checks against them); without it the
# records above stand in.  The reference pins torchrl 0.1.1 (README.md:59): spec class names of that release first.
try:  # pragma: no cover - not installed in the build image
    from torchrl.envs import EnvBase as TorchrlEnvBase  # type: ignore
    import torchrl.data as _trd  # type: ignore
    _RealComposite = getattr(_trd, "CompositeSpec", None) or getattr(_trd, "Composite")
    _RealUnbounded = getattr(_trd, "UnboundedContinuousTensorSpec", None) or getattr(_trd, "Unbounded")
    _RealBounded = getattr(_trd, "BoundedTensorSpec", None) or getattr(_trd, "Bounded")
    _RealDiscrete = getattr(_trd, "DiscreteTensorSpec", None) or getattr(_trd, "Categorical")
    USING_REAL_TORCHRL = _RealTensorDict is not None
except Exception:  # noqa: BLE001
    TorchrlEnvBase = None
    USING_REAL_TORCHRL = False


def unbounded_spec(shape, device=None, dtype=torch.float32):
    if USING_REAL_TORCHRL:
        return _RealUnbounded(shape, device=device, dtype=dtype)
    return TensorSpec(shape, dtype=dtype, device=device)


def bounded_spec(low, high, shape, device=None):
    if USING_REAL_TORCHRL:
        return _RealBounded(low, high, shape, device=device)
    return TensorSpec(shape, device=device, low=low, high=high)


def bool_spec(shape, device=None):
    if USING_REAL_TORCHRL:
        return _RealDiscrete(2, shape, dtype=torch.bool, device=device)
    return TensorSpec(shape, dtype=torch.bool, device=device)


def composite_spec(d, shape=()):
    """`shape` = the batch part every entry starts with: torchrl's EnvBase setters reject a composite whose shape differs from the env's
    batch_size (the reference builds per-env specs and `.expand(num_envs)`s them, hideandseek.py:327-433 — same result)."""
    if USING_REAL_TORCHRL:
        return _RealComposite({k: (composite_spec(v, shape) if isinstance(v, dict) and not isinstance(v, _RealComposite) else v) for k, v in d.items()},
                              shape=torch.Size(shape))
    return CompositeSpec({k: (composite_spec(v, shape) if isinstance(v, dict) and not isinstance(v, CompositeSpec) else v) for k, v in d.items()}, shape)


def spec_tree(spec):
    """{key: {...} | [shape, dtype]} of a (real or stand-in) spec tree — what the manifest test compares."""
    if isinstance(spec, (dict, CompositeSpec)) or (USING_REAL_TORCHRL and isinstance(spec, _RealComposite)):
        return {k: spec_tree(v) for k, v in spec.items()}
    return [list(spec.shape), str(spec.dtype).replace("torch.", "")]
