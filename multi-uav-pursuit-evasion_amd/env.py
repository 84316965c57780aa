"""`HideAndSeek(cfg, headless)` — the reference's environment class on top of the HIP step.

Mirror of the torchrl `EnvBase` surface the reference's caller uses
(omni_drones/envs/isaac_env.py:47-57,210-259, omni_drones/envs/hide_and_seek/hideandseek.py:183-433,
scripts/train.py:110-205): constructor `Env(cfg, headless)`, `REGISTRY`, `reset()`, `step(td)`,
`set_seed()`, `_reset/_step/_set_seed`, spec trees, `agent_spec["drone"]`, `num_envs`,
`max_episode_length`, `dt`, `progress_buf`, `stats`, `info`, `drone.params`, `drone.n`.

Where the action transform runs (`config.resolve_action_input`): with `task.action_transform: none`
(cfg/task/HideAndSeek_hip.yaml) the body-rate PID transform (omni_drones/utils/torchrl/transforms.py:404-459) is fused
into the step kernel and the env takes the RAW policy action (`action_input: policy`).  With the reference's own task file
(`action_transform: PIDrate`, cfg/task/HideAndSeek.yaml:16) scripts/train.py:165-171 puts the torch controller in front; the env
then takes its rotor commands and starts at `_pre_sim_step` (hideandseek.py:725-744), as the reference's env does
(`action_input: motor`) — the controller never runs twice.

Every tensor returned is a view of a persistent device buffer that the next `step()` overwrites
in place — the same aliasing contract as the reference (`hideandseek.py:901-902`).
There is no CPU path: constructing the env without the built HIP library or a GPU raises.
"""
import ctypes as C
from dataclasses import dataclass
from types import SimpleNamespace

import torch

from . import abi
from .config import CRAZYFLIE, resolve_hns_cfg
from .tensordict_shim import (USING_REAL_TENSORDICT, USING_REAL_TORCHRL, TensorDict, TorchrlEnvBase, bool_spec, bounded_spec,
                              composite_spec, unbounded_spec)


@dataclass
class AgentSpec:
    """omni_drones/utils/torchrl/env.py::AgentSpec (name, n and the four keys)."""
    name: str
    n: int
    observation_key: tuple = ("agents", "observation")
    action_key: tuple = ("agents", "action")
    reward_key: tuple = ("agents", "reward")
    state_key: tuple = ("agents", "state")
    _env: object = None

    @property
    def observation_spec(self):
        return self._env.observation_spec[self.observation_key]

    @property
    def action_spec(self):
        return self._env.action_spec[self.action_key]

    @property
    def reward_spec(self):
        return self._env.reward_spec[self.reward_key]

    @property
    def state_spec(self):
        return self._env.observation_spec[self.state_key]


class HnsError(RuntimeError):
    pass


class _LazyState(TensorDict):
    """`agents.state` whose `state_drones` entry is assembled in torch only if somebody reads it
    (the kernel skips that output unless `algo.critic_input: state`).  The entry is ONE persistent
    `[E,A,D]` buffer, refilled in place the first time it is read after each `step()` / `reset()` —
    the same aliasing contract as every other leaf (a view of a buffer the next step overwrites)."""

    def __init__(self, env, source, batch_size):
        super().__init__(source, batch_size)
        object.__setattr__(self, "_hns_env", env)
        object.__setattr__(self, "_filled_at", -1)

    def _refresh(self):
        env = self._hns_env
        if self._filled_at != env._state_version:
            buf = env._lazy_state_drones()
            if not dict.__contains__(self, "state_drones"):
                dict.__setitem__(self, "state_drones", buf)
            object.__setattr__(self, "_filled_at", env._state_version)

    def __getitem__(self, key):
        if key == "state_drones":
            self._refresh()
        return super().__getitem__(key)

    def keys(self, *a, **k):
        self._refresh()
        return super().keys(*a, **k)

    def items(self):
        self._refresh()
        return super().items()

    def values(self):
        self._refresh()
        return super().values()

    def _map(self, fn, batch_size=None):
        self._refresh()
        return super()._map(fn, batch_size)


try:                                           # the current stream's handle without building a torch.cuda.Stream object (~1.5 us per step)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:                         # older / newer torch without the private accessor
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream


class _PlainEnvBase:
    """What `torchrl.envs.EnvBase` gives the caller (`reset` / `step` / `set_seed` around `_reset` / `_step` /
    `_set_seed`, isaac_env.py:47-57) for installations without torchrl — this build image has none."""

    def __init__(self, device, batch_size, run_type_checks=False):
        self.device = torch.device(device)
        self.batch_size = torch.Size(batch_size)
        self.training = True

    def set_seed(self, seed=-1):
        self._set_seed(seed)
        return seed

    def reset(self, tensordict=None, **kwargs):
        td = self._reset(tensordict, **kwargs)
        if "done" not in td.keys():
            td.set("done", torch.zeros(self.batch_size[0], 1, dtype=torch.bool, device=self.device))
        return td

    def step(self, tensordict):
        out = self._step(tensordict)
        # EnvBase.step: tensordict.update(tensordict_out) — `next` (replaced as a whole: the same persistent tree every step) and the
        # transform's keys (merged into what the caller's tensordict already holds under `stats` / `info`).  The returned tree and its
        # leaves are persistent (views of the buffers the kernel rewrites in place), so a tensordict that already went through this
        # merge and still holds that tree needs nothing (a collector that reuses its tensordicts, return_same_td: ~5 us of Python per step)
        if isinstance(tensordict, dict) and getattr(tensordict, "_hns_merged", None) is out and self._still_merged(tensordict, out):
            return tensordict
        if not isinstance(tensordict, dict):      # a tensordict type that is not the dict-based shim (real tensordict without torchrl): plain update
            tensordict.update(out)
            return tensordict
        for k, v in dict.items(out):
            cur = tensordict.get(k)
            if cur is v:
                continue
            if k == "next" or cur is None or not isinstance(v, dict):
                tensordict.set(k, v)
            else:
                cur.update(v)
        try:
            tensordict._hns_merged = out
        except AttributeError:                    # a foreign tensordict type: merged every step
            pass
        return tensordict

    @staticmethod
    def _still_merged(tensordict, out):
        """Every entry of the returned tree is still the object the caller's tensordict holds (nobody replaced `next`, `stats`, ... since)."""
        for k, v in dict.items(out):
            cur = dict.get(tensordict, k)
            if cur is v:
                continue
            if k == "next" or not isinstance(v, dict) or not isinstance(cur, dict):
                return False
            for kk, vv in dict.items(v):
                if dict.get(cur, kk) is not vv:
                    return False
        return True

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)


def rasterise_top_down(pts, num_agents, num_targets, arena_size, max_height, cylinder_size, n=128):
    """`[n, n, 3]` uint8 top-down picture of one env: `pts` = `[A + NT + C, 3]` positions (pursuers, evader(s), cylinder slots).  Row 0 is +y, column 0 is -x;
    the frame spans 1.15 arena radii each way.  Arena disc (hideandseek.py:1094-1103: radius arena_size), active cylinders (z > 0; inactive slots sit at z = -20,
    :686-689), evader(s) in red (the reference's sphere has r = 0.05, :544-565), pursuers in blue — brighter the higher they fly."""
    import numpy as np
    pts = np.asarray(pts, dtype=np.float32)
    A, NT = int(num_agents), int(num_targets)
    R, H = float(arena_size), float(max_height)
    half = 1.15 * R                                          # metres from the frame's centre to its edge
    ax = (np.arange(n, dtype=np.float32) + 0.5) / n * 2.0 * half - half
    X, Y = np.meshgrid(ax, -ax)                              # row 0 = +y (image convention: y up)
    img = np.empty((n, n, 3), np.uint8)
    img[:] = (24, 24, 28)
    img[X * X + Y * Y <= R * R] = (58, 60, 66)

    def disc(p, radius, colour):
        img[(X - p[0]) ** 2 + (Y - p[1]) ** 2 <= radius * radius] = colour

    for c in pts[A + NT:]:
        if c[2] > 0.0:
            disc(c, cylinder_size, (150, 150, 150))
    px = 2.0 * half / n
    for k in range(NT):
        disc(pts[A + k], max(0.05, 1.5 * px), (230, 60, 50))
    for i in range(A):
        shade = int(120 + 135 * min(max(float(pts[i, 2]) / max(H, 1e-6), 0.0), 1.0))
        disc(pts[i], max(0.04, 1.5 * px), (40, shade, 255))
    return img


# With torchrl importable the class IS a torchrl `EnvBase` subclass, constructed the way `IsaacEnv` constructs itself
# (isaac_env.py:54-57: device, batch_size = [num_envs], run_type_checks = False), so `TransformedEnv(base_env, ...)`,
# the `SyncDataCollector` and MAPPO (scripts/train.py:165-205) take it as they take the reference's env.
_EnvBase = TorchrlEnvBase if USING_REAL_TORCHRL else _PlainEnvBase


class HideAndSeek(_EnvBase):
    REGISTRY = {}

    def __init__(self, cfg, headless=True, env_index_offset=0, write_critic_state=None):
        device = torch.device(cfg.sim.get("device", "cuda:0"))
        if device.type != "cuda":
            raise HnsError("HideAndSeek runs on an AMD GPU only (cfg.sim.device must be a cuda/hip device)")
        if not torch.cuda.is_available():
            raise HnsError("no GPU visible: the HIP step has no CPU fallback")
        super().__init__(device=device, batch_size=[int(cfg.env.num_envs)], run_type_checks=False)
        self._dev_index = device.index if device.index is not None else torch.cuda.current_device()
        self.cfg = cfg
        self.headless = headless
        self._lib = abi.load_library()
        if write_critic_state is None:
            # the centralised-critic state [E,A,20] is extra HBM traffic that only `critic_input: state`
            # consumes (reference cfg/algo/mappo.yaml:18 defaults to `obs`); when it is not written by
            # the kernel it is assembled lazily in torch on access (same values)
            write_critic_state = str(cfg.algo.get("critic_input", "obs")) == "state"
        self.write_critic_state = bool(write_critic_state) or USING_REAL_TENSORDICT   # lazy state needs the shim
        write_critic_state = self.write_critic_state
        self.num_envs = int(cfg.env.num_envs)
        self.max_episode_length = int(cfg.env.max_episode_length)
        self.dt = float(cfg.sim.dt)
        self.hcfg = resolve_hns_cfg(cfg, env_index_offset=env_index_offset, write_critic_state=write_critic_state)
        self.num_agents = self.hcfg.num_agents
        self.num_cylinders = self.hcfg.num_cylinders
        self.obs_max_cylinder = self.hcfg.obs_max_cylinder
        self.v_prey = float(self.hcfg.v_prey)
        self._update_epoch = 0                                   # `update_epoch` is a property: assigning it moves the smoothness schedule
        self.seed = 0
        self._render = not headless
        # transforms.py:456-457: `ctbr` and `target_rate` on the input tensordict — extra outputs, written only when asked for
        self.publish_ctbr = bool(int(cfg.task.get("publish_ctbr", 0)))
        # "policy": ("agents","action") is the raw policy output, A1/A2 run inside hns_step.  "motor": the caller's controller transform ran in front
        # (scripts/train.py:165-171) and left rotor commands there (transforms.py:455-456); hns_step starts at _pre_sim_step (include/hns.h: hns_action_input)
        self.action_input = "motor" if int(self.hcfg.action_input) == abi.HNS_ACTION_MOTOR else "policy"
        self._motor = self.action_input == "motor"
        if self._motor:
            self.publish_ctbr = False            # `ctbr` / `target_rate` are the transform's keys then (transforms.py:456-457)
        E, A, Cn, K = self.num_envs, self.num_agents, self.num_cylinders, self.obs_max_cylinder

        torch.cuda.set_device(self.device)
        self._bufs = {}
        self.num_targets = 2 if int(self.hcfg.num_targets) == 2 else 1
        for name, (shape, dt) in abi.buffer_shapes(E, A, Cn, K, self.num_targets, publish_ctbr=self.publish_ctbr).items():
            n = 1
            for s in shape:
                n *= s
            self._bufs[name] = torch.zeros(max(n, 1), dtype=getattr(torch, dt), device=self.device)[:n].view(shape)
        self._hbuf = abi.HnsBuffers()
        for name in abi.BUFFER_FIELDS:
            t = self._bufs.get(name)
            setattr(self._hbuf, name, t.data_ptr() if (t is not None and t.numel()) else None)
        if not write_critic_state:
            self._hbuf.state_drones = None
        # task.pid_reset = reference: `reset_pid = tensordict['done']` (transforms.py:449-454).  The step reads the byte at its very beginning
        # and writes `done` at its very end, so the input IS the done buffer: what the previous step (or a reset, which clears it) left there
        # is what the root `done` of the stepped tensordict holds in the collector's / rollout's loop; a tensordict whose root `done` is
        # another tensor is copied in first (`_step`)
        self.pid_reset_reference = int(self.hcfg.pid_reset_on_reset) == 0 and not self._motor     # (motor: the caller's controller owns that state)
        self._hbuf.reset_pid = self._bufs["done"].data_ptr() if self.pid_reset_reference else None
        self._env = C.c_void_p()
        self._check(self._lib.hns_create(C.byref(self.hcfg), C.byref(self._env)), "hns_create")
        self._check(self._lib.hns_bind(self._env, C.byref(self._hbuf)), "hns_bind")

        b = self._bufs
        self.progress_buf = b["progress"]
        self._tensordict = TensorDict({"progress": self.progress_buf}, self.batch_size)
        self.stats = TensorDict({k: b["stats"][i].unsqueeze(-1) for i, k in enumerate(abi.STAT_NAMES)}, self.batch_size)
        self.info = TensorDict({"drone_state": b["drone_state"], "prev_action": b["prev_action"]}, self.batch_size)
        self.drone = SimpleNamespace(n=A, params=CRAZYFLIE, throttle=b["throttle"], name="crazyflie",
                                     MASS_0=torch.tensor([CRAZYFLIE["mass"]]), num_rotors=4)
        # algo.use_TP_net=1 (reference default): the trajectory predictor runs between the step kernel and
        # the consumer (tp_net.py); the env owns the module, the learner trains it (scripts/train.py:180)
        self.use_TP_net = int(cfg.algo.get("use_TP_net", 0))
        self.TP = None
        if self.use_TP_net:
            from .tp_net import TPNet
            t = cfg.task
            self.tp_future_step = int(t.get("future_predcition_step", 5))
            self.tp_history_step = int(t.get("history_step", 10))
            self.use_obstacles = int(self.hcfg.tp_use_obstacles)                                # the frame also holds the cylinders
            self.tp_frame_dim = abi.tp_frame_dim(A, self.hcfg.num_cylinders, self.use_obstacles)
            self.TP = TPNet(self.tp_frame_dim, 3 * self.tp_future_step, self.tp_future_step,
                            int(t.get("window_step", 1))).to(self.device)                       # hideandseek.py:315-318
            self._tp_bufs = {}
            for name, (shape, dt) in abi.tp_buffer_shapes(E, A, self.tp_history_step, self.tp_future_step, self.tp_frame_dim, self.num_targets).items():
                if name not in abi.TP_WEIGHT_FIELDS:
                    self._tp_bufs[name] = torch.zeros(shape, dtype=getattr(torch, dt), device=self.device)
            self._tp_weight_ptrs = None
            self._tp_weight_versions = None
            self._tp_filled = False
        # Step + predictor as two half-batches on two streams (round 4): envs are independent, so the halves [0, E/2) and
        # [E/2, E) are stepped by two more handles over slices of the SAME buffers (pointers offset, `stats` addressed with the whole
        # batch's row stride), one on the caller's stream and one on a side stream, forked and joined with events inside `_step`.
        # OFF by default (task.tp_overlap: 1 turns it on): two free-running streams gain 7 % (118 -> 110 us per 65 536-env step,
        # tools/tp_overlap_lab.py), but `step()` must join them before it returns — its outputs feed the policy on the caller's stream — and
        # with a fork and a join per step the pair measures SLOWER than the whole batch on one stream (131.5 against 113.2 us, bench.py
        # `tp_mode`, round 4; 130.2 against 94.7 us at the end of round 6, tools/ab_env.py task.tp_overlap=0 task.tp_overlap=1 65536 --tp): kept for consumers that can take the halves un-joined (asynchronous collectors), bit-identical either way.
        self._halves = None
        ov = cfg.task.get("tp_overlap", 0)
        if self.use_TP_net and E % 128 == 0 and (ov == 1 or ov == "1"):
            self._make_halves()
        self._set_specs()
        self.success_rate_fn = None      # optional callable(env) -> success rate over the WHOLE (sharded) batch
        self._since_full_reset = 0
        self._needs_reset = True
        self._next_cache = None
        self._state_version = 0          # bumped by every step()/reset(): lazily assembled entries refill once per version
        self._state_buf = None
        self._action_shape = torch.Size([self.num_envs, self.num_agents, 4])
        self._done_ptr = self._bufs["done"].data_ptr()

    def _make_halves(self):
        E, h = self.num_envs, self.num_envs // 2
        self._side_stream = torch.cuda.Stream(self.device)
        self._ev_fork, self._ev_join = torch.cuda.Event(), torch.cuda.Event()
        halves = []
        for i in range(2):
            cfg_i = self.hcfg.copy()
            cfg_i.num_envs = h
            cfg_i.env_index_offset = int(self.hcfg.env_index_offset) + i * h
            cfg_i.stats_stride = E
            hb = abi.HnsBuffers()
            for name in abi.BUFFER_FIELDS:
                if name == "reset_pid":
                    continue
                t = self._bufs.get(name)
                ptr = getattr(self._hbuf, name)
                if t is None or not ptr:
                    setattr(hb, name, None)
                elif name == "stats":
                    setattr(hb, name, t[:, i * h:].data_ptr())           # columns [i h, (i + 1) h) of every row
                elif name == "nonfinite":
                    setattr(hb, name, t.data_ptr())                       # one sticky word for the whole batch
                else:
                    setattr(hb, name, t[i * h:].data_ptr())
            hb.reset_pid = self._bufs["done"][i * h:].data_ptr() if self.pid_reset_reference else None
            env_i = C.c_void_p()
            self._check(self._lib.hns_create(C.byref(cfg_i), C.byref(env_i)), "hns_create (half batch)")
            self._check(self._lib.hns_bind(env_i, C.byref(hb)), "hns_bind (half batch)")
            halves.append(SimpleNamespace(env=env_i, cfg=cfg_i, hbuf=hb, first=i * h, tp_bound=None))
        self._halves = halves

    def _bind_tp_halves(self, ws):
        """hns_tp_bind of the two half-batch handles: the learner's parameter tensors and the shared operand image, the unit-major buffers offset."""
        NT, h = self.num_targets, self.num_envs // 2
        for i, hv in enumerate(self._halves):
            tb = abi.HnsTpBuffers()
            for f, w in zip(abi.TP_WEIGHT_FIELDS, ws):
                setattr(tb, f, w.data_ptr())
            for name, t in self._tp_bufs.items():
                per_unit = name in ("history", "pred", "groundtruth", "tp_done")
                setattr(tb, name, t.data_ptr() if name == "packed" else t[i * h * (NT if per_unit else 1):].data_ptr())
            if not self.write_critic_state:
                tb.state_drones = None
            self._check(self._lib.hns_tp_bind(hv.env, C.byref(tb), self.tp_history_step, self.tp_future_step), "hns_tp_bind (half batch)")

    # ---- registry (isaac_env.py:154-161) ----------------------------------------------------------
    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        HideAndSeek.REGISTRY[cls.__name__] = cls
        HideAndSeek.REGISTRY[cls.__name__.lower()] = cls

    def _check(self, rc, what):
        if rc != 0:
            raise HnsError(f"{what} failed ({rc}): {self._lib.hns_last_error().decode()}")

    # ---- specs (hideandseek.py:327-433) ----------------------------------------------------------------------------
    def _set_specs(self):
        A, K, E, dev = self.num_agents, self.obs_max_cylinder, self.num_envs, self.device
        t = self.cfg.task
        F = int(t.get("future_predcition_step", 5))
        NT = self.num_targets
        D = abi.self_dim(NT) + (3 * F * NT if self.use_TP_net else 0)               # 20 or 35 (two evaders: 24 or 24 + 6F)
        # key order as the reference declares it (hideandseek.py:338-352: state_self, state_others, cylinders): the attention encoder builds its
        # token sequence in spec-key order with the first key as the query (learning/modules/networks.py:250-298)
        obs = {"state_self": unbounded_spec((E, A, 1, D), dev)}
        if A > 1:
            obs["state_others"] = unbounded_spec((E, A, A - 1, 3), dev)
        obs["cylinders"] = unbounded_spec((E, A, K, 5), dev)
        # (`state.cylinders` is declared (k, 5) although the tensor the reference returns is [E, A, k, 5], :887)
        state = {"state_drones": unbounded_spec((E, A, D), dev), "cylinders": unbounded_spec((E, K, 5), dev)}
        # the TP entry is part of the spec whether or not the predictor is used (:358-374)
        frame = abi.tp_frame_dim(A, self.num_cylinders, int(t.get("use_obstacles", 0)))
        tp = {"TP_input": unbounded_spec((E, int(t.get("history_step", 10)), frame) if NT == 1 else (E, NT, int(t.get("history_step", 10)), frame), dev),
              "TP_groundtruth": unbounded_spec((E, 1, 3) if NT == 1 else (E, NT, 3), dev), "TP_done": unbounded_spec((E, 1, 3), dev)}
        self.observation_spec = composite_spec({
            "agents": {"observation": obs, "state": state, "TP": tp},
            "stats": {k: unbounded_spec((E, 1), dev) for k in abi.STAT_NAMES},
            "info": {"drone_state": unbounded_spec((E, A, 13), dev), "prev_action": bounded_spec(-1.0, 1.0, (E, A, 4), dev)}}, (E,))
        self.action_spec = composite_spec({"agents": {"action": bounded_spec(-1.0, 1.0, (E, A, 4), dev)}}, (E,))
        self.reward_spec = composite_spec({"agents": {"reward": unbounded_spec((E, A, 1), dev)}}, (E,))
        if not USING_REAL_TORCHRL:                                # torchrl derives these two itself (input_spec["_action_spec"], a bool done_spec)
            self.done_spec = bool_spec((E, 1), dev)
            self.input_spec = composite_spec({"_action_spec": self.action_spec}, (E,))
        self.agent_spec = {"drone": AgentSpec("drone", A, _env=self)}

    # ---- besides reset / step / set_seed / train / eval, which the base class provides --------------------------------
    def _set_seed(self, seed=-1):
        """isaac_env.py:256-259: seeds torch; here it also keys the reset Philox stream."""
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        torch.manual_seed(int(seed))
        self._check(self._lib.hns_set_reset_epoch(self._env, 0), "hns_set_reset_epoch")

    @property
    def reset_epoch(self):
        return int(self._lib.hns_get_reset_epoch(self._env))

    def enable_render(self, enable=True):
        if not isinstance(enable, bool) and not callable(enable):                   # isaac_env.py:321-327: a bool or a callable(substep)
            raise TypeError("enable_render must be a bool or callable.")
        self._render = enable
        return bool(enable)

    def render(self, mode="human"):
        """`mode="rgb_array"`: an `[H, W, 3]` uint8 top-down frame of ONE env (`task.render_env`, default 0; `task.render_size`, default 128
        pixels) drawn from the bound buffers — arena disc, active cylinders, pursuers (brighter with height), evader(s) — which is what
        `evaluate()` stacks and transposes (scripts/train.py:220-223,251-254).  The reference's frame is the Isaac viewport (isaac_env.py:261-280);
        a viewport is out of scope (SURVEY §2 #17,#19), the frame's type and layout are not.  `mode="human"` (a window) returns None.
        Off the hot path: one read-back of that env's ~(3 A + 3 NT + 3 C) floats per call."""
        if mode != "rgb_array":
            if mode == "human":
                return None
            raise NotImplementedError(f"render mode {mode!r} (the reference knows 'human' and 'rgb_array', isaac_env.py:261-280)")
        if self._render is False:
            raise RuntimeError(f"Cannot render '{mode}' while rendering is disabled: call enable_render(True) first (isaac_env.py:333-338).")
        t = self.cfg.task
        e = min(max(int(t.get("render_env", 0)), 0), self.num_envs - 1)
        b = self._bufs
        # one device gather -> one copy to the host: [A + NT + C, 3]
        pts = torch.cat([b["drone_state"][e, :, 0:3], b["target_pos"][e].reshape(-1, 3), b["cylinders"][e].reshape(-1, 3)], dim=0).cpu().numpy()
        return rasterise_top_down(pts, self.num_agents, self.num_targets, float(self.hcfg.arena_size), float(self.hcfg.max_height),
                                  float(self.hcfg.cylinder_size), int(t.get("render_size", 128)))

    def to(self, device):
        if torch.device(device) != self.device:                  # isaac_env.py:300-305
            raise RuntimeError(f"Cannot move HideAndSeek on {self.device} to a different device {device} once it's initialized.")
        return self

    def close(self):
        for hv in (getattr(self, "_halves", None) or []):
            self._lib.hns_destroy(hv.env)
        self._halves = None
        if getattr(self, "_env", None):
            self._lib.hns_destroy(self._env)
            self._env = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- isaac_env.py:210-225 -----------------------------------------------------------------------------
    def _reset(self, tensordict=None, **kwargs):
        mask_t = None
        if tensordict is not None and "_reset" in tensordict.keys():     # (tensordict 0.1.x: get() without a default raises on a missing key)
            mask_t = tensordict.get("_reset").reshape(self.num_envs).to(torch.uint8).contiguous()
        last_stats = self._clone_stats()
        ptr = C.c_void_p(mask_t.data_ptr()) if mask_t is not None else None
        self._check(self._lib.hns_reset(self._env, ptr, C.c_uint64(self.seed), self._stream()), "hns_reset")
        self._reset_mask_keepalive = mask_t
        self._state_version += 1
        self._note_reset(mask_t)
        if self._needs_reset and mask_t is None:
            # first reset of this env: run the torch side of a MASKED reset once (mask conversion, the read-back of max(progress)) — the first
            # launch of a torch kernel in a process loads its code object (15-20 ms for this handful on a fresh box), better paid here than
            # at the first episode boundary of a rollout
            torch.zeros(self.num_envs, 1, dtype=torch.bool, device=self.device).reshape(self.num_envs).to(torch.uint8).contiguous()
            int(self.progress_buf.max().item())
        self._needs_reset = False
        if self.use_TP_net:
            self._tp_observe()
        td = self._obs_tensordict()
        if not self.training:
            td = self._fresh_obs(td)              # eval mode: new tensors, as in `_fresh_step_output`
        td.set("stats", last_stats)
        td.set("truncated", (self.progress_buf > self.max_episode_length).unsqueeze(1))
        return td

    def _clone_stats(self, skip=()):
        """`self.stats.clone()` (isaac_env.py:216: the statistics as they were BEFORE the reset) as ONE copy: the 24 statistics are rows of one
        buffer (24 small copy kernels otherwise — 0.1-0.2 ms of launches at every episode boundary).  Entries somebody added to `stats` are
        cloned one by one, except those in `skip` (a subclass that clones its own rows in one copy as well)."""
        base = self._bufs["stats"].clone()
        td = TensorDict({k: base[i].unsqueeze(-1) for i, k in enumerate(abi.STAT_NAMES)}, self.batch_size)
        for k in self.stats.keys():
            if k not in abi.STAT_NAMES and k not in skip:
                td.set(k, self.stats[k].clone())
        return td

    def _note_reset(self, mask_t):
        """Host mirror of "steps since the oldest running episode began": no env can be `done` before it reaches
        max_episode_length, so the per-step host checks (curriculum, generator) stay off until then.  A masked reset
        costs ONE read-back of max(progress) per reset call (once per episode), never one per step — and none at all while
        nothing consults the mirror (the evader-speed curriculum at its cap, no task generator): the counter then simply keeps
        running, an over-estimate that could only make those checks start early.  The read-back is a host sync: at an episode
        boundary it drains the queue of steps the host had run ahead by and leaves the device idle until the host has caught
        up — 0.2-0.3 ms per boundary at 65 536 envs (tools/lab/r04_batches_1_76.txt, batch 73)."""
        if mask_t is None:
            self._since_full_reset = 0
        elif self._episode_mirror_needed():
            self._since_full_reset = int(self.progress_buf.max().item())

    def _episode_mirror_needed(self):
        return self.v_prey < 1.3 - 1e-6

    # ---- transforms.py:425-459 + isaac_env.py:231-240 ------------------------------------------------
    def _step(self, tensordict):
        if self._needs_reset:
            raise HnsError("step() called before reset()")
        action = tensordict[("agents", "action")]
        if action.dtype != torch.float32 or not action.is_contiguous():
            action = action.float().contiguous()
        if action.shape != self._action_shape:
            raise ValueError(f"action shape {tuple(action.shape)} != {tuple(self._action_shape)}")
        if self._motor:
            self._take_transform_keys(tensordict)
        elif self.pid_reset_reference:
            # the incoming root `done` is the controller's reset_pid; the env's own done buffer (what `next.done` of the last step aliases,
            # cleared by reset for the envs it resets) already is that input unless the caller's tensordict carries another tensor
            d = tensordict.get("done", None)
            if d is not None and d.data_ptr() != self._done_ptr:
                self._bufs["done"].copy_(d.reshape(self.num_envs).to(torch.uint8))
        if self._halves is not None:
            self._step_halves(action)
        else:
            rc = self._lib.hns_step(self._env, action.data_ptr(), _raw_stream(self._dev_index))
            if rc != 0:
                self._check(rc, "hns_step")
        self._action_keepalive = action
        self._since_full_reset += 1
        self._state_version += 1
        b = self._bufs
        # hideandseek.py:1012-1015 — evader-speed curriculum; v_prey starts at its 1.3 cap with the
        # reference's defaults, in which case no host sync is ever needed
        if self.v_prey < 1.3 - 1e-6 and self._since_full_reset >= self.max_episode_length:
            done = b["done"].bool()
            # the reference averages over ALL envs (:1012-1015); a shard of a data-parallel run plugs the global rate in
            # through `success_rate_fn` (sharding.GlobalSuccessRate), otherwise the local batch is the whole batch
            rate = self.success_rate_fn(self) if self.success_rate_fn is not None else float(self.stats["success"].mean())
            if bool(done.any()) and rate >= 0.98:
                self.v_prey = min(1.3, self.v_prey + 0.05)
                for h in self._handles():
                    self._check(self._lib.hns_set_v_prey(h, C.c_float(self.v_prey)), "hns_set_v_prey")
        if self.use_TP_net and self._halves is None:
            self._tp_observe()
        if not self.training:
            return self._fresh_step_output()
        if self._next_cache is None:
            # every leaf is a view of a persistent buffer that the kernel just rewrote in place, so
            # the output tree is built once and handed out again (the reference's collector runs
            # with return_same_td=True, scripts/train.py:204)
            nxt = self._obs_tensordict()
            nxt.set(("agents", "reward"), b["reward"].unsqueeze(-1))
            nxt.set("done", b["done"].view(torch.bool).unsqueeze(-1))
            # beside `next`: what PIDRateController._inv_call leaves on the stepped tensordict (transforms.py:438-457; hideandseek.py:726-731
            # reads the first two back) — views of the buffers the kernel just updated.  They travel in the RETURNED tree, which the base
            # class merges into the caller's: torchrl locks the input tensordict while `_step` runs, new keys cannot be set on it.
            out = self._beside_next(nxt)
            self._next_cache = TensorDict(out, self.batch_size)
        return self._next_cache

    def _beside_next(self, nxt):
        """The tree `_step` returns: `next`, and — when the controller is fused into the step — the keys the reference's transform would have
        left on the stepped tensordict.  With `action_input: motor` that transform ran in the caller's wrapper and set them itself."""
        b = self._bufs
        if self._motor:
            return {"next": nxt}
        out = {"next": nxt, "stats": {"action_error_order1": b["action_error"]}, "info": {"prev_action": b["prev_action"]}}
        if self.publish_ctbr:
            out["ctbr"] = b["ctbr"]
            out["target_rate"] = b["target_rate"][..., :3]
        return out

    def _take_transform_keys(self, tensordict):
        """`action_input: motor` — hideandseek.py:729-731: `self.info["prev_action"] = tensordict[("info", "prev_action")]`,
        `self.action_error_order1 = tensordict[("stats", "action_error_order1")]`: what PIDRateController._inv_call left on the stepped tensordict
        (transforms.py:441-443) goes into the bound buffers (the kernel reads the action error for the statistics and the smoothness reward;
        `info.prev_action` is handed out with the next observation).  A tensordict without them fails here as it fails in the reference."""
        b = self._bufs
        try:
            prev = tensordict[("info", "prev_action")]
            aerr = tensordict[("stats", "action_error_order1")]
        except KeyError as e:
            raise KeyError(f"{e.args[0] if e.args else e}: with task.action_input = 'motor' (task.action_transform = "
                           f"{self.cfg.task.get('action_transform', None)!r}) the stepped tensordict must carry ('info', 'prev_action') and ('stats', "
                           "'action_error_order1') as PIDRateController._inv_call leaves them (omni_drones/utils/torchrl/transforms.py:441-443; read at "
                           "hideandseek.py:729-731).  To feed raw policy actions instead use cfg/task/HideAndSeek_hip.yaml (action_transform: none)") from None
        if prev.data_ptr() != b["prev_action"].data_ptr():
            b["prev_action"].copy_(prev.reshape(b["prev_action"].shape))
        if aerr.data_ptr() != b["action_error"].data_ptr():
            b["action_error"].copy_(aerr.reshape(b["action_error"].shape))

    def _fresh_step_output(self):
        """`env.eval()` (scripts/train.py:213-214 before `env.rollout(...)`, :225-233): the observation, state, predictor entries, reward and `done`
        are NEW tensors at every step, as the reference's are (torch.cat / stack results, hideandseek.py:856-917, 1056-1064) — `EnvBase.rollout`
        keeps the step outputs by reference (`tensordict.clone(False)`) and stacks them afterwards, which needs them to stay what they were.
        `stats` and `info` stay the persistent tensors they are in the reference as well (`"stats": self.stats`).  In train mode (the collector's
        path: it copies every step into its own storage) the leaves are zero-copy views of the buffers the next step rewrites."""
        b = self._bufs
        nxt = self._fresh_obs(self._obs_tensordict())
        nxt.set(("agents", "reward"), b["reward"].unsqueeze(-1).clone())
        nxt.set("done", b["done"].view(torch.bool).unsqueeze(-1).clone())
        return TensorDict(self._beside_next(nxt), self.batch_size)

    @staticmethod
    def _fresh_obs(td):
        agents = td.get("agents")
        for group in ("observation", "state", "TP"):
            g = agents.get(group, None)
            if g is not None:
                for k in list(g.keys()):
                    g.set(k, g.get(k).clone())
        return td

    def _step_halves(self, action):
        """hns_step + hns_tp_observe of [0, E/2) on the caller's stream and of [E/2, E) on the side stream, between a fork and a join event
        (legal inside a stream capture: the side stream joins the capture through the fork event)."""
        self._tp_sync_weights()
        main = torch.cuda.current_stream(self.device)
        side = self._side_stream
        lib, (ha, hb) = self._lib, self._halves
        fill = 0 if self._tp_filled else 1
        self._ev_fork.record(main)
        side.wait_event(self._ev_fork)
        a0 = action.data_ptr()
        rc = lib.hns_step(ha.env, a0, C.c_void_p(main.cuda_stream)) or lib.hns_tp_observe(ha.env, fill, C.c_void_p(main.cuda_stream))
        rc = rc or lib.hns_step(hb.env, a0 + hb.first * self.num_agents * 16, C.c_void_p(side.cuda_stream)) or lib.hns_tp_observe(hb.env, fill, C.c_void_p(side.cuda_stream))
        if rc != 0:
            self._check(rc, "hns_step / hns_tp_observe (half batch)")
        self._ev_join.record(side)
        main.wait_event(self._ev_join)
        self._tp_filled = True

    def _obs_tensordict(self):
        b = self._bufs
        if self.use_TP_net:
            tb = self._tp_bufs
            obs = {"state_self": tb["obs_self"].unsqueeze(2)}
            if self.num_agents > 1:
                obs["state_others"] = b["obs_others"]
            obs["cylinders"] = b["obs_cylinders"]
            if self.write_critic_state:
                state = {"state_drones": tb["state_drones"], "cylinders": b["obs_cylinders"]}
            else:
                state = _LazyState(self, {"cylinders": b["obs_cylinders"]}, self.batch_size)
            tp = {"TP_input": tb["history"], "TP_groundtruth": tb["groundtruth"], "TP_done": tb["tp_done"].view(torch.bool).unsqueeze(-1)}
            if self.num_targets == 2:            # extension: one window / ground truth per (env, evader); the done flag is the env's
                E = self.num_envs
                tp = {"TP_input": tb["history"].view(E, 2, *tb["history"].shape[1:]), "TP_groundtruth": tb["groundtruth"].view(E, 2, 3),
                      "TP_done": tb["tp_done"].view(torch.bool).view(E, 2)[:, :1]}
            return TensorDict({"agents": {"observation": obs, "state": state, "TP": tp}, "stats": self.stats, "info": self.info},
                              self.batch_size)
        obs = {"state_self": b["obs_self"].unsqueeze(2)}
        if self.num_agents > 1:
            obs["state_others"] = b["obs_others"]
        obs["cylinders"] = b["obs_cylinders"]
        if self.write_critic_state:
            state = {"state_drones": b["state_drones"], "cylinders": b["obs_cylinders"]}
        else:
            state = _LazyState(self, {"cylinders": b["obs_cylinders"]}, self.batch_size)
        return TensorDict({"agents": {"observation": obs, "state": state}, "stats": self.stats, "info": self.info},
                          self.batch_size)

    def _lazy_state_drones(self):
        """hideandseek.py:871-886: state_self with the UNMASKED relative position of the evader."""
        b = self._bufs
        rest = self._tp_bufs["obs_self"] if self.use_TP_net else b["obs_self"]       # :873-880 / :881-886
        if self._state_buf is None:
            self._state_buf = torch.empty_like(rest)
        if self.num_targets == 2:                                                    # extension: both relative positions unmasked
            rpos = b["drone_state"][..., None, 0:3] - b["target_pos"].unsqueeze(1)   # [E,A,2,3]
            R = 3 * self.tp_future_step if self.use_TP_net else 0                    # evader 0's predictions sit behind its relative position
            return torch.cat([rpos[:, :, 0], rest[..., 3:R + 20], rpos[:, :, 1], rest[..., R + 23:]], dim=-1, out=self._state_buf)
        rpos = b["drone_state"][..., 0:3] - b["target_pos"].unsqueeze(1)
        return torch.cat([rpos, rest[..., 3:]], dim=-1, out=self._state_buf)

    def _tp_observe(self):
        """The TP branch of `_compute_state_and_obs` (hideandseek.py:805-854) on the device: frame
        append, TP_net forward on the matrix cores, 35-value rows (hns_tp_observe).  `self.TP`'s
        parameters are re-packed into the operand image only when their version counters moved
        (optimiser step / load_state_dict), re-bound only if the learner swapped the tensors."""
        self._tp_sync_weights()
        rc = self._lib.hns_tp_observe(self._env, 0 if self._tp_filled else 1, self._stream())
        if rc != 0:
            self._check(rc, "hns_tp_observe")
        self._tp_filled = True                      # the window is never reset per env (hideandseek.py:825-830)

    def _tp_sync_weights(self):
        """(Re-)bind / re-pack the predictor's parameters when the learner swapped or updated them (see `_tp_observe`)."""
        tp = self.TP            # attribute lookups, not state_dict(): this runs every step
        ws = (tp.lstm.weight_ih_l0, tp.lstm.weight_hh_l0, tp.lstm.bias_ih_l0, tp.lstm.bias_hh_l0, tp.fc.weight, tp.fc.bias)
        ptrs = tuple(w.data_ptr() for w in ws)
        versions = tuple(w._version for w in ws)         # bumped by every in-place update (optimiser step, copy_)
        if ptrs != self._tp_weight_ptrs:
            tb = abi.HnsTpBuffers()
            for f, w in zip(abi.TP_WEIGHT_FIELDS, ws):
                if w.dtype != torch.float32 or not w.is_contiguous() or w.device != self.device:
                    raise HnsError(f"TP_net parameter {abi.TP_STATE_DICT_KEYS[f]} must be a contiguous fp32 tensor on {self.device}")
                setattr(tb, f, w.data_ptr())
            for name, t in self._tp_bufs.items():
                setattr(tb, name, t.data_ptr())
            if not self.write_critic_state:
                tb.state_drones = None
            self._check(self._lib.hns_tp_bind(self._env, C.byref(tb), self.tp_history_step, self.tp_future_step), "hns_tp_bind")
            if self._halves is not None:
                self._bind_tp_halves(ws)
            self._tp_weight_ptrs, self._tp_weight_versions = ptrs, versions
            self._tp_refresh_all()
        elif versions != self._tp_weight_versions:
            self._tp_refresh_all()
            self._tp_weight_versions = versions

    def _tp_refresh_all(self):
        """One re-pack of the operand image on the caller's stream (the handles share it); every handle is told its image is current."""
        for h in [self._env] + [hv.env for hv in (self._halves or [])]:
            self._check(self._lib.hns_tp_refresh(h, self._stream()), "hns_tp_refresh")

    def refresh_tp_weights(self):
        """Re-pack the predictor's parameters now.  `_tp_observe` notices optimiser steps and `load_state_dict` through
        the tensors' version counters; writes through `.data` do not move them — call this after such an update."""
        if self.use_TP_net and self._tp_weight_ptrs is not None:
            self._tp_refresh_all()

    # ---- schedule hooks -------------------------------------------------------------------------------------------
    @property
    def update_epoch(self):
        return self._update_epoch

    @update_epoch.setter
    def update_epoch(self, epoch):
        """`base_env.update_epoch = i` (scripts/train_deploy.py:270) is read at every step by the reference's reward (hideandseek.py:988-989:
        smoothness_coef = min(max, init + smooth_lr * update_epoch)); here the assignment itself pushes the coefficient to the kernel's
        configuration — one stream-ordered 4-byte copy, and only when the coefficient actually moved."""
        self._update_epoch = int(epoch)
        t = self.cfg.task
        init_s = float(t.get("init_smoothness_coef", t.get("smoothness_coef", 0.0)))            # (the envgen YAML names it smoothness_coef)
        coef = min(float(t.get("max_smoothness_coef", 5.0)), init_s + float(t.get("smooth_lr", 0.0)) * self._update_epoch)
        if getattr(self, "_env", None) and coef != getattr(self, "_smoothness_coef_pushed", None):
            for h in self._handles():
                self._check(self._lib.hns_set_smoothness_coef(h, C.c_float(coef)), "hns_set_smoothness_coef")
            self._smoothness_coef_pushed = coef

    def set_update_epoch(self, epoch):
        """Method form of the assignment above (kept for callers of rounds 1-4)."""
        self.update_epoch = epoch

    def _handles(self):
        return [self._env] + [hv.env for hv in (self._halves or [])]

    def _timed_handle(self):
        return self._halves[0].env if self._halves is not None else self._env      # (halves: the launches of [0, E/2))

    def enable_kernel_timing(self, on=True):
        self._check(self._lib.hns_enable_timing(self._timed_handle(), int(on)), "hns_enable_timing")

    def kernel_ms(self):
        n = C.c_int(0)
        ms = float(self._lib.hns_step_kernel_ms(self._timed_handle(), C.byref(n)))
        return ms, n.value

    @property
    def step_mapping(self):
        """'tile' or 'small': which mapping of the step kernel serves this env (hns_step_mapping; csrc/hns_step_small_kernel.h)."""
        return "small" if self._lib.hns_step_mapping(self._env) == 1 else "tile"

    def region_begin(self):
        """One start event on the stream the steps are launched on (hns_region_begin); `region_end` records the stop event,
        `region_ms` waits for it.  Nothing is added to the launches in between."""
        self._check(self._lib.hns_region_begin(self._env, _raw_stream(self._dev_index)), "hns_region_begin")

    def region_end(self):
        self._check(self._lib.hns_region_end(self._env, _raw_stream(self._dev_index)), "hns_region_end")

    def region_ms(self):
        return float(self._lib.hns_region_ms(self._env))

    def clock_probe(self, ticks=2000):
        """Enqueue one clock probe (hns_clock_probe: one wave for `ticks` x 10 ns) on the current stream; returns the device tensor [shader cycles, 100 MHz ticks] —
        read it after a synchronisation with `clock_mhz(t)`.  Nothing is read back here."""
        out = torch.zeros(2, dtype=torch.int64, device=self.device)
        self._check(self._lib.hns_clock_probe(out.data_ptr(), int(ticks), _raw_stream(self._dev_index)), "hns_clock_probe")
        return out

    @staticmethod
    def clock_mhz(probe):
        c, r = (int(x) for x in probe.tolist())
        return 100.0 * c / r if r > 0 else None

    def device_copy_GBs(self, mbytes=256, reps=20):
        """The box's achievable HBM rate with the library's float4 copy kernel (hns_copy_f4): read + written bytes per second."""
        n = mbytes * 1024 * 1024
        src = torch.empty(n // 4, dtype=torch.float32, device=self.device).normal_()
        dst = torch.empty_like(src)
        st = _raw_stream(self._dev_index)
        for _ in range(3):
            self._check(self._lib.hns_copy_f4(dst.data_ptr(), src.data_ptr(), n, st), "hns_copy_f4")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            self._lib.hns_copy_f4(dst.data_ptr(), src.data_ptr(), n, st)
        e1.record()
        torch.cuda.synchronize(self.device)
        assert torch.equal(dst, src)
        return reps * 2 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9

    # ---- helpers for tests / tools --------------------------------------------------------------------------------
    def rand_step_input(self, action=None):
        if action is None:
            action = torch.randn(self.num_envs, self.num_agents, 4, device=self.device)
        return TensorDict({"agents": {"action": action}}, self.batch_size)

    def export_state(self):
        torch.cuda.synchronize(self.device)
        return {k: v.detach().cpu().numpy().copy() for k, v in self._bufs.items()}

    def import_state(self, arrays, check=False):
        for k, v in arrays.items():
            self._bufs[k].copy_(torch.as_tensor(v).to(self.device).view(self._bufs[k].shape))
        if check and not all(bool(torch.isfinite(self._bufs[k]).all()) for k in arrays if self._bufs[k].dtype.is_floating_point):
            raise HnsError("import_state: non-finite values in the imported state")
        # the line-of-sight column of pid_last_rate is derived from the positions just written (include/hns.h)
        self._check(self._lib.hns_refresh_derived_state(self._env, self._stream()), "hns_refresh_derived_state")
        self._needs_reset = False
        self._state_version += 1

    def raycast(self, num_rays=16, max_range=2.0, out=None):
        """Extension (not in the reference): planar ray-fan ranges [E,A,num_rays] on the current state."""
        if out is None:
            out = torch.empty(self.num_envs, self.num_agents, num_rays, device=self.device)
        self._check(self._lib.hns_raycast(self._env, int(num_rays), C.c_float(max_range), C.c_void_p(out.data_ptr()),
                                          self._stream()), "hns_raycast")
        return out

    # ---- checkpoint / resume of the env state (the reference checkpoints the policy only, train.py:288-292) ----
    def load_policy_checkpoint(self, checkpoint):
        """Take the trajectory predictor's parameters from a checkpoint of the reference's MAPPO policy
        (`MAPPOPolicy.state_dict()`, learning/mappo.py:477-484: {"TP", "critic", "actor_params", "value_normalizer"};
        written by scripts/train.py:292,318 with torch.save).  `checkpoint` is that dict or a path to it."""
        if not self.use_TP_net:
            raise HnsError("load_policy_checkpoint: algo.use_TP_net is off, there is no predictor to load")
        if not isinstance(checkpoint, dict):
            checkpoint = torch.load(checkpoint, map_location="cpu")
        self.TP.load_state_dict(checkpoint["TP"])        # in-place copy: the version counters move, hns_tp_refresh follows

    def save_state(self, path):
        """Snapshot every bound buffer + the host-side counters to an .npz file."""
        import numpy as np
        meta = np.array([self.seed & 0x7FFFFFFFFFFFFFFF, self.reset_epoch, self._since_full_reset, self.update_epoch], dtype=np.int64)
        extra = {}
        if self.use_TP_net:                         # the predictor's window is state too
            extra = {"_tp_history": self._tp_bufs["history"].cpu().numpy(), "_tp_filled": np.int64(self._tp_filled)}
        np.savez_compressed(path, _meta=meta, _v_prey=np.float64(self.v_prey), **extra, **self.export_state())

    def load_state(self, path):
        import numpy as np
        z = np.load(path)
        # derived / optional outputs may be absent from a snapshot written by an older build or without `publish_ctbr`: the sticky
        # failure word restarts at zero, the transform's extra outputs are rewritten by the next step
        optional = ("nonfinite", "ctbr", "target_rate")
        fields = [k for k in self._bufs if k in z.files or k not in optional]
        for k in fields:
            if k not in z.files:
                raise KeyError(f"snapshot has no field {k}")
            if tuple(z[k].shape) != tuple(self._bufs[k].shape):
                raise ValueError(f"snapshot field {k} has shape {z[k].shape}, env expects {tuple(self._bufs[k].shape)}")
        for k in optional:
            if k in self._bufs and k not in z.files:
                self._bufs[k].zero_()
        self.import_state({k: z[k] for k in fields})
        self.seed, epoch, self._since_full_reset, self.update_epoch = (int(x) for x in z["_meta"])
        self._check(self._lib.hns_set_reset_epoch(self._env, C.c_uint32(epoch)), "hns_set_reset_epoch")
        self.v_prey = float(z["_v_prey"])
        for h in self._handles():
            self._check(self._lib.hns_set_v_prey(h, C.c_float(self.v_prey)), "hns_set_v_prey")
        self.set_update_epoch(self.update_epoch)
        if self.use_TP_net and "_tp_history" in z.files:
            self._tp_bufs["history"].copy_(torch.from_numpy(z["_tp_history"]))
            self._tp_filled = bool(int(z["_tp_filled"]))

    def check_finite(self, clear=False, deep=False):
        """Failure detection: True iff no STEP since the word was last cleared produced a non-finite pursuer state, evader
        position or reward.  The step kernel ORs one sticky device word (hns_buffers.nonfinite); this reads that word —
        one 4-byte read-back, no reduction over the buffers.  Non-finite values that enter another way (import_state / load_state,
        a reset, the observation buffers) are caught one step later at the earliest — they propagate into the state the step checks —
        or at once with `deep=True`, which reduces every float buffer (what the soak tools and `import_state(..., check=True)` use)."""
        word = int(self._bufs["nonfinite"].item())
        if clear:
            self._bufs["nonfinite"].zero_()
        ok = word == 0
        if deep:
            ok = ok and all(bool(torch.isfinite(v).all()) for k, v in self._bufs.items() if v.dtype.is_floating_point)
        return ok

    def nonfinite_bits(self):
        """The raw word: bit 0 pursuer state, bit 1 evader position, bit 2 reward."""
        return int(self._bufs["nonfinite"].item())


HideAndSeek.REGISTRY["HideAndSeek"] = HideAndSeek
HideAndSeek.REGISTRY["hideandseek"] = HideAndSeek
HideAndSeek.REGISTRY["HideAndSeek_hip"] = HideAndSeek        # cfg/task/HideAndSeek_hip.yaml (action_transform: none)
