"""`Hover(cfg, headless)` — the reference's single-drone hover task (BASELINE config 1, plumbing
scale) over `hns_hover_step` / `hns_hover_reset`.

Mirrors omni_drones/envs/single/hover.py:40-523 with the default observation options
(`omega/motor/add_noise/latency/action_noise = false`, `time_encoding = true`): observation
`[E,1,20] = [target_pos - pos, quat, linvel, heading, up, t x4]`, the 39 statistics of
hover.py:238-278, reward of hover.py:439-476.  As in `HideAndSeek`, the body-rate PID transform is
fused into the step, so `step()` takes the raw policy action.
"""
import ctypes as C

import torch

from . import abi
from .config import CRAZYFLIE, resolve_hover_cfg
from .env import AgentSpec, HideAndSeek, HnsError
from .tensordict_shim import CompositeSpec, TensorDict, TensorSpec


class Hover:
    def __init__(self, cfg, headless=True, env_index_offset=0):
        self.cfg = cfg
        self.device = torch.device(cfg.sim.get("device", "cuda:0"))
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise HnsError("Hover runs on an AMD GPU only: the HIP step has no CPU fallback")
        self._lib = abi.load_library()
        self.num_envs = int(cfg.env.num_envs)
        self.max_episode_length = int(cfg.env.max_episode_length)
        self.dt = float(cfg.sim.dt)
        self.batch_size = torch.Size([self.num_envs])
        self.hcfg, self.hover_cfg = resolve_hover_cfg(cfg, env_index_offset)
        self.seed, self.reset_epoch, self.training = 0, 0, True
        E = self.num_envs
        torch.cuda.set_device(self.device)
        self._bufs = {k: torch.zeros(shape, dtype=getattr(torch, dt), device=self.device)
                      for k, (shape, dt) in abi.hover_buffer_shapes(E).items()}
        self._hbuf = abi.HnsHoverBuffers()
        for k in abi.HOVER_BUFFER_FIELDS:
            setattr(self._hbuf, k, self._bufs[k].data_ptr())
        b = self._bufs
        self.progress_buf = b["progress"]
        self.stats = TensorDict({k: b["stats"][i].unsqueeze(-1) for i, k in enumerate(abi.HOVER_STAT_NAMES)}, self.batch_size)
        self.info = TensorDict({"drone_state": b["drone_state"], "prev_action": b["prev_action"]}, self.batch_size)
        self.drone = type("Drone", (), {"n": 1, "params": CRAZYFLIE, "throttle": b["throttle"], "num_rotors": 4})()
        self.observation_spec = CompositeSpec({
            "agents": CompositeSpec({"observation": TensorSpec((1, abi.HNS_SELF_DIM)), "intrinsics": TensorSpec((1, 0))}),
            "stats": CompositeSpec({k: TensorSpec((1,)) for k in abi.HOVER_STAT_NAMES}),
            "info": CompositeSpec({"drone_state": TensorSpec((1, 13)), "prev_action": TensorSpec((1, 4))})}).expand(E).to(self.device)
        self.action_spec = CompositeSpec({"agents": CompositeSpec({"action": TensorSpec((1, 4), low=-1.0, high=1.0)})}).expand(E).to(self.device)
        self.reward_spec = CompositeSpec({"agents": CompositeSpec({"reward": TensorSpec((1, 1))})}).expand(E).to(self.device)
        self.agent_spec = {"drone": AgentSpec("drone", 1, state_key=("agents", "intrinsics"), _env=self)}
        self._needs_reset = True

    def _check(self, rc, what):
        if rc != 0:
            raise HnsError(f"{what} failed ({rc}): {self._lib.hns_last_error().decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_seed(self, seed=-1):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.reset_epoch = 0
        torch.manual_seed(int(seed))
        return seed

    def to(self, device):
        if torch.device(device) != self.device:
            raise RuntimeError(f"Cannot move Hover on {self.device} to a different device {device} once it's initialized.")
        return self

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def close(self):
        pass

    def _obs_tensordict(self):
        b = self._bufs
        return TensorDict({"agents": {"observation": b["obs"], "intrinsics": torch.zeros(self.num_envs, 1, 0, device=self.device)},
                           "stats": self.stats, "info": self.info}, self.batch_size)

    def reset(self, tensordict=None, **kwargs):
        mask_t = None
        if tensordict is not None and tensordict.get("_reset") is not None:
            mask_t = tensordict.get("_reset").reshape(self.num_envs).to(torch.uint8).contiguous()
        last_stats = self.stats.clone()
        self._check(self._lib.hns_hover_reset(C.byref(self.hcfg), C.byref(self.hover_cfg), C.byref(self._hbuf),
                                              C.c_void_p(mask_t.data_ptr()) if mask_t is not None else None,
                                              C.c_uint64(self.seed), C.c_uint32(self.reset_epoch), self._stream()), "hns_hover_reset")
        self.reset_epoch += 1
        self._keep = mask_t
        self._needs_reset = False
        td = self._obs_tensordict()
        td.set("stats", last_stats)
        td.set("truncated", (self.progress_buf > self.max_episode_length).unsqueeze(1))
        td.set("done", torch.zeros(self.num_envs, 1, dtype=torch.bool, device=self.device))
        return td

    def step(self, tensordict):
        if self._needs_reset:
            raise HnsError("step() called before reset()")
        action = tensordict[("agents", "action")]
        if action.dtype != torch.float32 or not action.is_contiguous():
            action = action.float().contiguous()
        if tuple(action.shape) != (self.num_envs, 1, 4):
            raise ValueError(f"action shape {tuple(action.shape)} != {(self.num_envs, 1, 4)}")
        self._check(self._lib.hns_hover_step(C.byref(self.hcfg), C.byref(self.hover_cfg), C.byref(self._hbuf),
                                             C.c_void_p(action.data_ptr()), self._stream()), "hns_hover_step")
        self._keep_action = action
        b = self._bufs
        nxt = self._obs_tensordict()
        nxt.set(("agents", "reward"), b["reward"].unsqueeze(-1))
        nxt.set("done", b["done"].view(torch.bool).unsqueeze(-1))
        tensordict.set("next", nxt)
        return tensordict

    def rand_step_input(self, action=None):
        if action is None:
            action = torch.randn(self.num_envs, 1, 4, device=self.device)
        return TensorDict({"agents": {"action": action}}, self.batch_size)

    def export_state(self):
        torch.cuda.synchronize(self.device)
        return {k: v.detach().cpu().numpy().copy() for k, v in self._bufs.items()}

    def import_state(self, arrays):
        for k, v in arrays.items():
            self._bufs[k].copy_(torch.as_tensor(v).to(self.device).view(self._bufs[k].shape))
        self._needs_reset = False


HideAndSeek.REGISTRY["Hover"] = Hover
HideAndSeek.REGISTRY["hover"] = Hover
