"""Stand-in for tensordict 0.1.2 (see ../README.md): strict where the real one is strict."""
import torch

_NO_DEFAULT = object()


class TensorDictBase:
    pass


def _is_key(k):
    return isinstance(k, str) or (isinstance(k, tuple) and len(k) > 0 and all(isinstance(x, str) for x in k))


class TensorDict(TensorDictBase):
    def __init__(self, source=None, batch_size=None, device=None):
        if batch_size is None:
            raise TypeError("batch_size is required")
        self._batch_size = torch.Size(batch_size)
        self._device = torch.device(device) if device is not None else None
        self._d = {}
        self._locked = False
        for k, v in (source or {}).items():
            self.set(k, v)

    # ---- properties -------------------------------------------------------------------------------------------
    @property
    def batch_size(self):
        return self._batch_size

    @property
    def shape(self):
        return self._batch_size

    @property
    def device(self):
        return self._device

    @property
    def is_locked(self):
        return self._locked

    def lock_(self):
        self._locked = True
        for v in self._d.values():
            if isinstance(v, TensorDict):
                v.lock_()
        return self

    def unlock_(self):
        self._locked = False
        for v in self._d.values():
            if isinstance(v, TensorDict):
                v.unlock_()
        return self

    def numel(self):
        n = 1
        for s in self._batch_size:
            n *= s
        return n

    # ---- leaves -------------------------------------------------------------------------------------------------
    def _check_leaf(self, key, v):
        nb = len(self._batch_size)
        if isinstance(v, TensorDictBase):
            if tuple(v.batch_size[:nb]) != tuple(self._batch_size):
                raise RuntimeError(f"batch dimension mismatch, got self.batch_size={self._batch_size} and value.batch_size={v.batch_size} for key {key}")
            return v
        if isinstance(v, dict):
            return TensorDict(v, self._batch_size, self._device)
        if not isinstance(v, torch.Tensor):
            raise TypeError(f"tensordict values must be tensors, tensordicts or dicts, got {type(v)} for key {key}")
        if tuple(v.shape[:nb]) != tuple(self._batch_size):
            raise RuntimeError(f"batch dimension mismatch, got self.batch_size={self._batch_size} and value.shape[:self.batch_dims]={v.shape[:nb]} with value of shape {tuple(v.shape)} (key {key})")
        return v

    def set(self, key, value, inplace=False):
        if not _is_key(key):
            raise TypeError(f"invalid key {key!r}")
        if isinstance(key, tuple) and len(key) == 1:
            key = key[0]
        if isinstance(key, tuple):
            head, rest = key[0], key[1:]
            if head not in self._d:
                if self._locked:
                    raise RuntimeError("Cannot modify locked TensorDict. For in-place modification, consider using the `set_()` method and make sure the key is present.")
                self._d[head] = TensorDict({}, self._batch_size, self._device)
            sub = self._d[head]
            if not isinstance(sub, TensorDict):
                raise KeyError(f"{head} is a leaf, cannot set {key}")
            sub.set(rest if len(rest) > 1 else rest[0], value, inplace)
            return self
        if self._locked and key not in self._d:
            raise RuntimeError("Cannot modify locked TensorDict. For in-place modification, consider using the `set_()` method and make sure the key is present.")
        self._d[key] = self._check_leaf(key, value)
        return self

    def get(self, key, default=_NO_DEFAULT):
        try:
            cur = self
            for k in ((key,) if isinstance(key, str) else key):
                if not isinstance(cur, TensorDict):
                    raise KeyError(key)
                cur = cur._d[k]
            return cur
        except KeyError:
            if default is _NO_DEFAULT:
                raise KeyError(f'key "{key}" not found in TensorDict with keys {sorted(map(str, self._d))}') from None
            return default

    def __getitem__(self, key):
        if _is_key(key):
            return self.get(key)
        probe = torch.empty(self._batch_size, device="meta")[key]
        out = TensorDict({}, probe.shape, self._device)
        for k, v in self._d.items():
            out._d[k] = v[key]
        return out

    def __setitem__(self, key, value):
        if _is_key(key):
            self.set(key, value)
            return
        for k, v in self._d.items():
            v[key] = value.get(k) if isinstance(value, TensorDictBase) else value

    def __contains__(self, key):
        raise NotImplementedError("TensorDict does not support membership checks with the `in` keyword; use `key in tensordict.keys()`")

    def keys(self, include_nested=False, leaves_only=False):
        out = []
        for k, v in self._d.items():
            if isinstance(v, TensorDict):
                if not leaves_only:
                    out.append(k)
                if include_nested:
                    out.extend((k, *(s if isinstance(s, tuple) else (s,))) for s in v.keys(True, leaves_only))
            else:
                out.append(k)
        return out

    def items(self, include_nested=False, leaves_only=False):
        return [(k, self.get(k)) for k in self.keys(include_nested, leaves_only)]

    def values(self):
        return list(self._d.values())

    def update(self, other, inplace=False):
        items = other.items() if not isinstance(other, TensorDictBase) else [(k, other._d[k]) for k in other._d]
        for k, v in items:
            cur = self._d.get(k)
            if isinstance(cur, TensorDict) and isinstance(v, (TensorDictBase, dict)):
                cur.update(v)
            else:
                self.set(k, v)
        return self

    def _map(self, fn, batch_size=None):
        out = TensorDict({}, self._batch_size if batch_size is None else batch_size, self._device)
        for k, v in self._d.items():
            out._d[k] = v._map(fn, batch_size) if isinstance(v, TensorDict) else fn(v)
        return out

    def clone(self, recurse=True):
        return self._map((lambda t: t.clone()) if recurse else (lambda t: t))

    def to(self, device):
        out = self._map(lambda t: t.to(device))
        out._device = torch.device(device)
        return out

    def cpu(self):
        return self.to("cpu")

    def select(self, *keys, inplace=False, strict=True):
        out = TensorDict({}, self._batch_size, self._device)
        for k in keys:
            v = self.get(k, None)
            if v is None:
                if strict:
                    raise KeyError(k)
                continue
            out.set(k, v)
        if inplace:
            self._d = out._d
            return self
        return out

    def exclude(self, *keys, inplace=False):
        out = self if inplace else self._map(lambda t: t)
        for k in keys:
            k = (k,) if isinstance(k, str) else k
            cur = out
            for kk in k[:-1]:
                cur = cur._d.get(kk)
                if cur is None:
                    break
            if isinstance(cur, TensorDict):
                cur._d.pop(k[-1], None)
        return out

    def __repr__(self):
        def shapes(d):
            return {k: (shapes(v) if isinstance(v, TensorDict) else tuple(v.shape)) for k, v in d._d.items()}
        return f"TensorDict(fake 0.1.2, batch_size={tuple(self._batch_size)}, fields={shapes(self)})"
