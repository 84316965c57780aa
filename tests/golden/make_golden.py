#!/usr/bin/env python3
"""Generate golden input/output vectors by EXECUTING the reference's own pure-torch code.

Runs only in the authoring container (needs /root/reference; never on the GPU box, never
from tests).  Nothing of the reference's source is written to the fixtures: they hold
inputs and the outputs the reference functions produced for them.

How the reference is executed without Isaac Sim / torchrl / tensordict:
  * `omni_drones/utils/torch.py`, `actuators/rotor_group.py` and
    `controllers/lee_position_controller.py` are loaded BY FILE PATH (a 1-symbol `tensordict`
    stub satisfies the controller's unused import).
  * Functions/methods of `envs/hide_and_seek/hideandseek.py`, `robots/drone/multirotor.py`
    and `utils/torchrl/transforms.py` (whose module tops import Isaac/torchrl) are
    extracted by AST *at run time* and exec'd against shim objects (tensor-backed views,
    dict-backed TensorDict).

Fixtures written (tests/golden/*.npz):
  g_utils      quat_rotate / quat_rotate_inverse / quat_axis / euler_to_quaternion / normalize
  g_rotor      RotorGroup.forward trajectories                      (rotor_group.py:55-71)
  g_pid        transforms.PIDRateController._inv_call + controllers.PIDRateController.forward
               sequences                                            (transforms.py:425-459,
                                                                     lee_position_controller.py:476-550)
  g_downwash   MultirotorBase.downwash summed over j                (multirotor.py:725-753)
  g_apply      MultirotorBase.apply_action                          (multirotor.py:466-508)
  g_blocked    is_line_blocked_by_cylinder                          (hideandseek.py:47-103)
  g_prey       _get_dummy_policy_prey + velocity line               (hideandseek.py:737-744,1067-1141)
  g_obs        get_state + _compute_state_and_obs (use_TP_net=0)    (multirotor.py:599-633, hideandseek.py:746-917)
  g_reward     _compute_reward_and_done incl. a done step           (hideandseek.py:919-1065)
  g_grid       continuous_to_grid / grid_to_continuous / set_outside_circle_to_one (hideandseek.py:121-181)
  g_episode_*  closed-loop episodes: reference functions for every stage + the build's
               integrator spec (A5, self-golden for that one stage), teacher-forcing states stored.
  g_genbuffer  the reference's GenBuffer class and the curriculum statements of `_compute_reward_and_done`, executed as written
               (hideandseek_envgen.py:209-377, :1241-1246, :1302-1336): init_easy_cases, the bounds samplenearby clips to, samplenearby outputs,
               two task batches through insert / insert_weights / update / statistics / R_min..R_max filter / insert_history (exact FPS, named start)
  g_learner_moments  MAPPOPolicy.train_op's advantage normalisation + ValueNorm1.update / normalize (mappo.py:391-402, valuenorm.py:83-98)
               on whole [E,T,A,1] tensors: what the data-parallel form (sharding.py) must reproduce from gathered moments
  g_episode_resetpid_*  the same loop with the controller's reset_pid = the root `done` of the stepped tensordict
               (transforms.py:449-454), across resets of the done envs (their deterministic effects restated, see the function)
"""
import ast
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch
import yaml
from functorch import vmap

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


# --------------------------------------------------------------------------------------
# loading helpers
# --------------------------------------------------------------------------------------
def load_by_path(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def install_stubs():
    # parent packages so that `from omni_drones.utils.torch import ...` resolves to the
    # file-path-loaded module without ever running omni_drones/__init__.py (imports Isaac)
    for name in ["omni_drones", "omni_drones.utils", "omni_drones.actuators", "omni_drones.controllers"]:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    td = types.ModuleType("tensordict")
    td.TensorDict = TensorDict
    sys.modules["tensordict"] = td


def extract_source(relpath, names, classname=None):
    """Return {name: compiled-source-text} for module-level functions or methods of a class."""
    src = open(os.path.join(REF, relpath)).read()
    tree = ast.parse(src)
    out = {}
    body = tree.body
    if classname is not None:
        body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == classname][0].body
    for node in body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            seg = ast.get_source_segment(src, node)
            # drop decorators (staticmethod) - we bind explicitly
            out[node.name] = (seg, node.col_offset)
    missing = set(names) - set(out)
    assert not missing, missing
    return out


def exec_functions(sources, namespace):
    import textwrap
    fns = {}
    for name, (seg, col) in sources.items():
        code = textwrap.dedent(" " * col + seg)
        exec(compile(code, f"<ref:{name}>", "exec"), namespace)
        fns[name] = namespace[name]
    return fns


# --------------------------------------------------------------------------------------
# shims
# --------------------------------------------------------------------------------------
class TensorDict(dict):
    """dict-backed stand-in for tensordict.TensorDict (nested dicts, tuple keys)."""

    def __init__(self, source=None, batch_size=None, device=None):
        super().__init__()
        self.batch_size = batch_size
        for k, v in (source or {}).items():
            self[k] = TensorDict(v, batch_size) if isinstance(v, dict) and not isinstance(v, TensorDict) else v

    def __getitem__(self, key):
        if isinstance(key, tuple):
            cur = self
            for k in key:
                cur = dict.__getitem__(cur, k)
            return cur
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def set(self, key, value):
        if isinstance(key, tuple):
            cur = self
            for k in key[:-1]:
                if k not in cur:
                    dict.__setitem__(cur, k, TensorDict({}, self.batch_size))
                cur = dict.__getitem__(cur, k)
            dict.__setitem__(cur, key[-1], value)
        else:
            dict.__setitem__(self, key, value)
        return self

    def update(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(dict.get(self, k), TensorDict):
                dict.__getitem__(self, k).update(v)          # nested merge, as tensordict does
            else:
                self[k] = TensorDict(v, self.batch_size) if isinstance(v, dict) and not isinstance(v, TensorDict) else v
        return self

    def clone(self):
        return TensorDict({k: v.clone() for k, v in self.items()}, self.batch_size)


class View:
    def __init__(self, pos, vel=None):
        self.pos = pos
        E, n = pos.shape[:2]
        self.rot = torch.zeros(E, n, 4)
        self.rot[..., 0] = 1
        self.vel = vel if vel is not None else torch.zeros(E, n, 6)

    def get_world_poses(self, clone=True):
        return self.pos.clone(), self.rot.clone()

    def get_velocities(self, clone=True):
        return self.vel.clone()

    def set_velocities(self, vel, env_ids=None):
        self.vel = vel.clone()


class Recorder:
    def __init__(self):
        self.calls = []

    def apply_forces_and_torques_at_pos(self, forces=None, torques=None, positions=None, is_global=False):
        self.calls.append(dict(forces=None if forces is None else forces.clone(),
                               torques=None if torques is None else torques.clone(),
                               positions=positions, is_global=is_global))


# --------------------------------------------------------------------------------------
# reference pieces
# --------------------------------------------------------------------------------------
install_stubs()
ref_torch = load_by_path("omni_drones.utils.torch", "omni_drones/utils/torch.py")
ref_rotor = load_by_path("omni_drones.actuators.rotor_group", "omni_drones/actuators/rotor_group.py")
ref_ctrl = load_by_path("omni_drones.controllers.lee_position_controller",
                        "omni_drones/controllers/lee_position_controller.py")
CF = yaml.safe_load(open(os.path.join(REF, "omni_drones/robots/assets/usd/crazyflie.yaml")))
TASK = yaml.safe_load(open(os.path.join(REF, "cfg/task/HideAndSeek.yaml")))
DT = 0.01

HNS = "omni_drones/envs/hide_and_seek/hideandseek.py"
hns_ns = dict(torch=torch, np=np, vmap=vmap, math=math, TensorDict=TensorDict,
              cpos=ref_torch.cpos, off_diag=ref_torch.off_diag, quat_axis=ref_torch.quat_axis,
              euler_to_quaternion=ref_torch.euler_to_quaternion, TensorDictBase=TensorDict)
hns_fn = exec_functions(extract_source(HNS, [
    "is_perpendicular_line_intersecting_segment", "is_line_blocked_by_cylinder",
    "grid_to_continuous", "continuous_to_grid", "set_outside_circle_to_one"]), hns_ns)
hns_m = exec_functions(extract_source(HNS, [
    "_get_dummy_policy_prey", "_compute_state_and_obs", "_compute_reward_and_done", "_pre_sim_step"],
    classname="HideAndSeek"), hns_ns)

MR = "omni_drones/robots/drone/multirotor.py"
mr_ns = dict(torch=torch, vmap=vmap, quat_axis=ref_torch.quat_axis, quat_rotate=ref_torch.quat_rotate,
             quat_rotate_inverse=ref_torch.quat_rotate_inverse, normalize=ref_torch.normalize,
             off_diag=ref_torch.off_diag)
mr_f = exec_functions(extract_source(MR, ["separation"]), mr_ns)
mr_m = exec_functions(extract_source(MR, ["apply_action", "get_state", "downwash"], classname="MultirotorBase"), mr_ns)
# `downwash` is a @staticmethod in the class body: the extracted segment starts at `def`
downwash = mr_ns["downwash"]

TR = "omni_drones/utils/torchrl/transforms.py"
tr_ns = dict(torch=torch, TensorDictBase=TensorDict)
tr_m = exec_functions(extract_source(TR, ["_inv_call"], classname="PIDRateController"), tr_ns)


class ShimTransform:
    """`self` for transforms.PIDRateController._inv_call (attributes set in its __init__, :405-418)."""

    def __init__(self):
        self.controller = ref_ctrl.PIDRateController(DT, 9.81, CF)
        self.action_key = ("agents", "action")
        self.target_clip = self.controller.target_clip
        self.max_thrust_ratio = self.controller.max_thrust_ratio
        self.fixed_yaw = self.controller.fixed_yaw

    def inv(self, td):
        return tr_m["_inv_call"](self, td)


class ShimDrone:
    """`self` for MultirotorBase.apply_action / get_state (attributes per multirotor.py:200-262)."""

    def __init__(self, E, A):
        self.shape = (E, A)
        self.n = A
        self.num_rotors = 4
        self.dt = DT
        self.params = CF
        self.rotor_module = ref_rotor.RotorGroup(CF["rotor_configuration"], dt=DT)
        rm = self.rotor_module
        # per-drone copies of the rotor parameters == rotor_params.expand(shape).clone() (:210)
        for name in ["KF", "KM", "throttle", "directions", "tau_up", "tau_down"]:
            p = getattr(rm, name)
            setattr(rm, name, torch.nn.Parameter(p.data.expand(E, A, 4).clone(), requires_grad=False))
        self.throttle = rm.throttle  # same storage, as in the reference (:216)
        self.KF = rm.KF
        self.rotor_params = None
        self.thrusts = torch.zeros(E, A, 4, 3)
        self.torques = torch.zeros(E, A, 3)
        self.forces = torch.zeros(E, A, 3)
        self.pos = torch.zeros(E, A, 3)
        self.rot = torch.zeros(E, A, 4)
        self.vel = self.vel_w = torch.zeros(E, A, 6)
        self.vel_b = torch.zeros(E, A, 6)
        self.heading = torch.zeros(E, A, 3)
        self.up = torch.zeros(E, A, 3)
        self.throttle_difference = torch.zeros(E, A)
        self.is_articulation = False
        self.rotor_joint_indices = None
        self.use_force_sensor = False
        self.mass = CF["mass"]
        self.masses = torch.ones(E, A, 1) * CF["mass"]
        self.drag_coef = torch.zeros(E, A, 1) * CF["drag_coef"]
        self.rotors_view = self
        self.base_link = Recorder()
        self.rotor_rec = Recorder()
        self.rotor_pos_offset = None
        # "physics" state the views return
        self.w_pos = torch.zeros(E, A, 3)
        self.w_rot = torch.zeros(E, A, 4)
        self.w_vel = torch.zeros(E, A, 6)
        self.downwash = downwash

    # the elementwise RotorGroup.forward on [E,A,4] parameters equals vmap(vmap(rotors)) over
    # per-drone parameter copies (multirotor.py:469-471); the reference's own forward() runs.
    def rotors(self, cmds, params):
        raise RuntimeError("replaced by the vmap shim")

    def get_world_poses(self, clone=True):
        return self.w_pos.clone(), self.w_rot.clone()

    def get_velocities(self, clone=True):
        return self.w_vel.clone()

    def set_state(self, pos, rot, vel):
        self.w_pos, self.w_rot, self.w_vel = pos.clone(), rot.clone(), vel.clone()

    def get_state(self):
        return mr_m["get_state"](self)

    def apply_action(self, cmds):
        # namespace-level vmap shim: `vmap(vmap(self.rotors, ...), ...)` (multirotor.py:469) becomes the
        # rotor module's own elementwise forward; every other vmap (downwash) is the real functorch.vmap
        real_vmap = vmap
        drone = self

        def rotor_call(cmds_, params_):
            return drone.rotor_module(cmds_)
        rotor_call._rotor = True

        def vmap_shim(f, *a, **k):
            if getattr(f, "_rotor", False):
                return f
            if getattr(f, "__self__", None) is drone and getattr(f, "__name__", "") == "rotors":
                return rotor_call
            return real_vmap(f, *a, **k)

        mr_ns["vmap"] = vmap_shim
        try:
            self.rotors_view = types.SimpleNamespace(
                get_world_poses=lambda: (None, self.w_rot.unsqueeze(2).expand(*self.shape, 4, 4).clone()),
                apply_forces_and_torques_at_pos=self.rotor_rec.apply_forces_and_torques_at_pos)
            out = mr_m["apply_action"](self, cmds)
        finally:
            mr_ns["vmap"] = real_vmap
        return out


class ShimEnv:
    """`self` for the HideAndSeek methods (attributes per hideandseek.py:236-325, 435-455)."""
    STATS = ["success", "collision", "blocked", "distance_reward", "distance_predicted_reward",
             "speed_reward", "collision_reward", "collision_wall", "collision_cylinder",
             "collision_drone", "detect_reward", "catch_reward", "smoothness_reward",
             "smoothness_mean", "smoothness_max", "first_capture_step", "sum_detect_step",
             "return", "action_error_order1_mean", "action_error_order1_max",
             "target_predicted_error", "distance_threshold_L", "out_of_arena", "smoothness_coef"]

    def __init__(self, E, A, C, task=None, max_len=800):
        t = dict(TASK)
        t.update(task or {})
        self.num_envs, self.num_agents, self.num_cylinders = E, A, C
        self.device = "cpu"
        self.batch_size = [E]
        self.max_episode_length = max_len
        self.cfg = types.SimpleNamespace(task=types.SimpleNamespace(v_drone=t["v_drone"]))
        self.drone = ShimDrone(E, A)
        self.target = View(torch.zeros(E, 1, 3))
        self.cylinders = View(torch.zeros(E, C, 3))
        self.env_ids = torch.arange(E)
        self.arena_size = t["arena_size"]
        self.max_height = t["max_height"]
        self.cylinder_size = t["cylinder"]["size"]
        self.cylinder_height = self.max_height
        self.obs_max_cylinder = t["cylinder"]["obs_max_cylinder"]
        self.drone_detect_radius = t["drone_detect_radius"]
        self.target_detect_radius = t["target_detect_radius"]
        self.catch_radius = t["catch_radius"]
        self.collision_radius = t["collision_radius"]
        self.v_prey = t["v_drone"] * t["v_prey"]
        self.catch_reward_coef = t["catch_reward_coef"]
        self.detect_reward_coef = t["detect_reward_coef"]
        self.collision_coef = t["collision_coef"]
        self.speed_coef = t["speed_coef"]
        self.dist_reward_coef = t["dist_reward_coef"]
        self.init_smoothness_coef = t["init_smoothness_coef"]
        self.max_smoothness_coef = t["max_smoothness_coef"]
        self.smooth_lr = t["smooth_lr"]
        self.update_epoch = 0
        self.use_eval = t["use_eval"]
        self.use_deployment = t["use_deployment"]
        self.use_TP_net = 0
        self.use_obstacles = t["use_obstacles"]
        self.time_encoding_dim = 4
        self.mask_value = -5
        self.progress_buf = torch.zeros(E)
        self.stats = TensorDict({k: torch.zeros(E, 1) for k in self.STATS}, [E])
        self.stats["first_capture_step"] = torch.ones(E, 1) * max_len
        self.info = TensorDict({"drone_state": torch.zeros(E, A, 13), "prev_action": torch.zeros(E, A, 4)}, [E])
        self.prev_actions = torch.zeros(E, A, 4)
        self.cylinders_mask = torch.zeros(E, C, dtype=torch.bool)

    def get_env_poses(self, poses):
        return poses

    def _should_render(self, substep):
        return False

    def _get_dummy_policy_prey(self):
        return hns_m["_get_dummy_policy_prey"](self)

    def _pre_sim_step(self, td):
        return hns_m["_pre_sim_step"](self, td)

    def _compute_state_and_obs(self):
        return hns_m["_compute_state_and_obs"](self)

    def _compute_reward_and_done(self):
        return hns_m["_compute_reward_and_done"](self)


# --------------------------------------------------------------------------------------
# random scene helpers
# --------------------------------------------------------------------------------------
def rand_quat(g, *shape, max_tilt=0.6):
    rpy = (torch.rand(*shape, 3, generator=g) * 2 - 1) * torch.tensor([max_tilt, max_tilt, math.pi])
    return ref_torch.euler_to_quaternion(rpy)


def rand_scene(g, E, A, C, n_active=None, spread=0.8):
    """Random but plausible rigid state inside the arena. Returns dict of tensors."""
    pos = (torch.rand(E, A, 3, generator=g) * 2 - 1) * torch.tensor([spread, spread, 0.0]) \
        + torch.tensor([0.0, 0.0, 0.2]) + torch.rand(E, A, 3, generator=g) * torch.tensor([0.0, 0.0, 0.9])
    rot = rand_quat(g, E, A)
    vel = torch.randn(E, A, 6, generator=g) * torch.tensor([0.5, 0.5, 0.3, 2.0, 2.0, 1.0])
    tpos = (torch.rand(E, 1, 3, generator=g) * 2 - 1) * torch.tensor([spread, spread, 0.0]) \
        + torch.tensor([0.0, 0.0, 0.1]) + torch.rand(E, 1, 3, generator=g) * torch.tensor([0.0, 0.0, 1.0])
    # cylinders on the 9x9 grid (cell 0.2) like the reference's reset, distinct cells
    cyl = torch.zeros(E, C, 3)
    for e in range(E):
        cells = torch.randperm(81, generator=g)[:C]
        cyl[e, :, 0] = ((cells // 9).float() - 4) * 0.2
        cyl[e, :, 1] = ((cells % 9).float() - 4) * 0.2
    cyl = cyl.clamp(-0.8, 0.8)
    cyl[..., 2] = 0.6
    if n_active is None:
        n_act = torch.randint(0, C + 1, (E, 1), generator=g) if C > 0 else torch.zeros(E, 1, dtype=torch.long)
    else:
        n_act = torch.full((E, 1), n_active)
    inactive = torch.arange(C).unsqueeze(0).expand(E, -1) >= n_act
    cyl[..., 2][inactive] = -20.0
    return dict(pos=pos, rot=rot, vel=vel, tpos=tpos, cyl=cyl)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KB  keys={len(out)}")


# --------------------------------------------------------------------------------------
# generators
# --------------------------------------------------------------------------------------
def gen_utils():
    g = torch.Generator().manual_seed(20240922)
    q = rand_quat(g, 256, max_tilt=1.5)
    v = torch.randn(256, 3, generator=g) * 3
    rpy = (torch.rand(256, 3, generator=g) * 2 - 1) * math.pi
    x = torch.randn(256, 3, generator=g)
    x[0] = 0.0
    save("g_utils", q=q, v=v, rpy=rpy, x=x,
         rotate=ref_torch.quat_rotate(q, v), rotate_inv=ref_torch.quat_rotate_inverse(q, v),
         axis0=ref_torch.quat_axis(q, 0), axis2=ref_torch.quat_axis(q, 2),
         e2q=ref_torch.euler_to_quaternion(rpy), normalize=ref_torch.normalize(x))


def gen_rotor():
    g = torch.Generator().manual_seed(20240923)
    E, A, T = 16, 3, 50
    d = ShimDrone(E, A)
    thr0 = torch.rand(E, A, 4, generator=g)
    d.rotor_module.throttle.data.copy_(thr0)
    cmds = (torch.rand(T, E, A, 4, generator=g) * 2 - 1) * 1.2
    cmds[10:20] = cmds[10:11]  # hold a command so the lag converges
    thrusts, moments, throttles = [], [], []
    for t in range(T):
        th, mo = d.rotor_module(cmds[t])
        thrusts.append(th.clone()); moments.append(mo.clone()); throttles.append(d.throttle.data.clone())
    save("g_rotor", throttle0=thr0, cmds=cmds, thrusts=torch.stack(thrusts), moments=torch.stack(moments),
         throttles=torch.stack(throttles), KF=d.rotor_module.KF.data[0, 0], KM=d.rotor_module.KM.data[0, 0])


def gen_pid():
    g = torch.Generator().manual_seed(20240924)
    E, A, T = 16, 3, 40
    tr = ShimTransform()
    action = torch.randn(T, E, A, 4, generator=g) * 1.5
    action[5] *= 20.0  # saturate tanh / thrust clamp
    rot = rand_quat(g, T, E, A)
    angvel = torch.randn(T, E, A, 3, generator=g) * torch.tensor([3.0, 3.0, 1.0])
    angvel[7] *= 30.0  # saturate the output clip
    done = torch.zeros(T, E, 1, dtype=torch.bool)
    done[12, ::2] = True
    done[25] = True
    prev = torch.zeros(E, A, 4)
    prev[..., 3] = 0.575
    prev0 = prev.clone()
    out = dict(cmds=[], ctbr=[], aerr=[], prev=[], target_rate=[], integ=[], last=[])
    for t in range(T):
        ds = torch.zeros(E, A, 13)
        ds[..., 3:7] = rot[t]
        ds[..., 10:13] = angvel[t]
        td = TensorDict({"agents": {"action": action[t].clone()},
                         "info": {"drone_state": ds, "prev_action": prev.clone()},
                         "stats": {}, "done": done[t]}, [E])
        td = tr.inv(td)
        prev = td[("info", "prev_action")].clone()
        out["cmds"].append(td[("agents", "action")].clone())
        out["ctbr"].append(td["ctbr"].clone())
        out["aerr"].append(td[("stats", "action_error_order1")].clone())
        out["prev"].append(prev.clone())
        out["target_rate"].append(td["target_rate"].clone())
        out["integ"].append(tr.controller.integ.clone().reshape(E, A, 3))
        out["last"].append(tr.controller.last_body_rate.clone().reshape(E, A, 3))
    save("g_pid", action=action, rot=rot, angvel=angvel, done=done, prev0=prev0,
         **{k: torch.stack(v) for k, v in out.items()})


def gen_downwash():
    arrs = {}
    for A in (2, 3, 6):
        g = torch.Generator().manual_seed(20240925 + A)
        E = 64
        pos = (torch.rand(E, A, 3, generator=g) * 2 - 1) * torch.tensor([0.3, 0.3, 0.5])
        # vertically stacked pair, and a z == 0 degeneracy (same height, level drones)
        pos[0, 1] = pos[0, 0] + torch.tensor([0.0, 0.0, -0.3])
        pos[1, 1] = pos[1, 0] + torch.tensor([0.2, 0.0, 0.0])
        rot = rand_quat(g, E, A, max_tilt=0.4)
        rot[0] = torch.tensor([1.0, 0, 0, 0]); rot[1] = torch.tensor([1.0, 0, 0, 0])
        tsum = torch.rand(E, A, generator=g) * 0.5
        tvec = torch.zeros(E, A, 3); tvec[..., 2] = tsum
        p1_t = ref_torch.quat_rotate(rot, tvec)
        f = vmap(downwash)(pos, pos, p1_t, kz=0.3).sum(-2)
        arrs.update({f"pos{A}": pos, f"rot{A}": rot, f"tsum{A}": tsum, f"f{A}": f})
    save("g_downwash", **arrs)


def gen_apply():
    arrs = {}
    for A in (1, 3):
        g = torch.Generator().manual_seed(20240930 + A)
        E = 32
        d = ShimDrone(E, A)
        s = rand_scene(g, E, A, 1)
        d.set_state(s["pos"], s["rot"], s["vel"])
        d.get_state()  # fills the pos/rot/vel caches apply_action reads (multirotor.py:600-610)
        thr0 = torch.rand(E, A, 4, generator=g)
        d.rotor_module.throttle.data.copy_(thr0)
        cmds = (torch.rand(E, A, 4, generator=g) * 2 - 1) * 1.1
        eff = d.apply_action(cmds)
        thrust_call = d.rotor_rec.calls[-1]
        base_call = d.base_link.calls[-1]
        assert thrust_call["is_global"] is False and base_call["is_global"] is True
        arrs.update({f"pos{A}": s["pos"], f"rot{A}": s["rot"], f"vel{A}": s["vel"], f"thr0_{A}": thr0,
                     f"cmds{A}": cmds, f"thr1_{A}": d.throttle.data.clone(),
                     f"rotor_force_local{A}": thrust_call["forces"].reshape(E, A, 4, 3),
                     f"base_force_world{A}": base_call["forces"].reshape(E, A, 3),
                     f"base_torque_world{A}": base_call["torques"].reshape(E, A, 3),
                     f"thr_diff{A}": d.throttle_difference.clone(), f"effort{A}": eff})
    save("g_apply", **arrs)


def gen_blocked():
    g = torch.Generator().manual_seed(20240940)
    E, A, C = 1024, 3, 8
    s = rand_scene(g, E, A, C)
    dp, tp, cy = s["pos"], s["tpos"], s["cyl"]
    dp[0, 0] = tp[0, 0]                      # degenerate drone == target
    # clear blocking: cylinder on the midpoint
    cy[1, 0, :2] = 0.5 * (dp[1, 0, :2] + tp[1, 0, :2]); cy[1, 0, 2] = 0.6
    # same but inactive
    cy[2, 0, :2] = 0.5 * (dp[2, 0, :2] + tp[2, 0, :2]); cy[2, 0, 2] = -20.0
    # cylinder behind the target on the line (t outside [0,1])
    cy[3, 0, :2] = tp[3, 0, :2] + 1.5 * (tp[3, 0, :2] - dp[3, 0, :2]); cy[3, 0, 2] = 0.6
    blocked = hns_fn["is_line_blocked_by_cylinder"](dp, tp, cy, 0.1)
    # float64 evaluation of the same formulas -> margin mask (which cases are safely away from a threshold)
    d64, t64, c64 = dp.double(), tp.double(), cy.double()
    diff = d64 - t64; diff2 = c64 - t64
    num = (diff[..., 0].unsqueeze(-1) * diff2[..., 1].unsqueeze(1) - diff[..., 1].unsqueeze(-1) * diff2[..., 0].unsqueeze(1)).abs()
    den = torch.sqrt(diff[..., 0] ** 2 + diff[..., 1] ** 2).unsqueeze(-1)
    dist = num / (den + 1e-5)
    dx = (t64[:, :, 0] - d64[:, :, 0]); dy = (t64[:, :, 1] - d64[:, :, 1])
    tt = ((c64[:, :, 0].unsqueeze(1) - d64[:, :, 0].unsqueeze(2)) * dx.unsqueeze(2)
          + (c64[:, :, 1].unsqueeze(1) - d64[:, :, 1].unsqueeze(2)) * dy.unsqueeze(2)) / (dx.unsqueeze(2) ** 2 + dy.unsqueeze(2) ** 2 + 1e-5)
    margin = torch.minimum((dist - 0.1).abs(), torch.minimum(tt.abs(), (tt - 1).abs()))
    safe = (margin > 1e-5).all(-1)
    save("g_blocked", drone_pos=dp, target_pos=tp, cyl=cy, blocked=blocked, safe=safe)


def gen_prey():
    arrs = {}
    for tag, (A, C, task) in {"a3c8": (3, 8, {}), "a3c5": (3, 5, {}),
                              "a6c16": (6, 16, {}), "a3c8_r": (3, 8, {"target_detect_radius": 0.6})}.items():
        g = torch.Generator().manual_seed(20240950 + len(tag) + A + C)
        E = 128
        env = ShimEnv(E, A, C, task)
        s = rand_scene(g, E, A, C)
        tp = s["tpos"]
        tp[0, 0] = torch.tensor([1.2, 0.3, 0.5])     # outside arena
        tp[1, 0, 2] = 1.35                           # above the ceiling
        tp[2, 0, 2] = -0.05                          # below ground
        tp[3, 0] = torch.tensor([0.0, 0.0, 0.6])     # at origin (direction 0/1e-5)
        env.drone.set_state(s["pos"], s["rot"], s["vel"])
        env.target.pos = tp
        env.cylinders.pos = s["cyl"]
        # stale-by-design mask: what the previous obs pass wrote (hideandseek.py:759,1134)
        env.cylinders_mask = s["cyl"][..., 2] < 0.0
        td = TensorDict({"agents": {"action": torch.zeros(E, A, 4)},
                         "info": {"prev_action": torch.zeros(E, A, 4)},
                         "stats": {"action_error_order1": torch.zeros(E, A)}}, [E])
        # only the prey part of _pre_sim_step matters here; apply_action runs but is ignored
        env.drone.get_state()
        force = env._get_dummy_policy_prey()
        ooa = env.stats["out_of_arena"].clone()
        env._pre_sim_step(td)
        arrs.update({f"{tag}_drone_pos": s["pos"], f"{tag}_target_pos": tp, f"{tag}_cyl": s["cyl"],
                     f"{tag}_force": force, f"{tag}_vel": env.target.vel[..., :3], f"{tag}_out_of_arena": ooa,
                     f"{tag}_detect_radius": np.float32(env.target_detect_radius)})
    save("g_prey", **arrs)


def _stats_arr(env):
    return torch.cat([env.stats[k] for k in ShimEnv.STATS], dim=1)


def gen_obs_reward():
    obs_arrs, rew_arrs = {}, {}
    cases = {"a3c8": (3, 8, {}, None), "a3c5": (3, 5, {}, None), "a3c5_none": (3, 5, {}, 0),
             "a6c16": (6, 16, {}, None), "a2c3": (2, 3, {}, 2),
             "a3c8_r": (3, 8, {"drone_detect_radius": 0.7, "use_deployment": 1, "init_smoothness_coef": 2.0}, None)}
    for tag, (A, C, task, n_active) in cases.items():
        g = torch.Generator().manual_seed(20240960 + 7 * A + C + len(tag))
        E = 96
        env = ShimEnv(E, A, C, task)
        s = rand_scene(g, E, A, C, n_active=n_active)
        # boundary-ish cases: capture, drone-drone collision, cylinder collision, wall, ceiling
        s["pos"][0, 0] = s["tpos"][0, 0] + torch.tensor([0.1, 0.05, 0.02])
        if A > 1:
            s["pos"][1, 1] = s["pos"][1, 0] + torch.tensor([0.05, 0.0, 0.02])
        s["pos"][2, 0, :2] = s["cyl"][2, 0, :2] + torch.tensor([0.12, 0.0]); s["pos"][2, 0, 2] = 0.6
        s["pos"][3, 0] = torch.tensor([0.85, 0.4, 0.5])
        s["pos"][4, 0, 2] = 1.25
        s["vel"][5, 0, :3] = torch.tensor([0.9, 0.6, 0.1])
        thr = torch.rand(E, A, 4, generator=g)
        env.drone.rotor_module.throttle.data.copy_(thr)
        env.drone.set_state(s["pos"], s["rot"], s["vel"])
        env.target.pos = s["tpos"]
        env.target.vel = torch.randn(E, 1, 6, generator=g)
        env.cylinders.pos = s["cyl"]
        progress = torch.randint(1, 799, (E,), generator=g).float()
        progress[E // 2:] = 800.0   # done rows -> stats division
        progress[-1] = 801.0
        env.progress_buf = progress.clone()
        stats0 = torch.rand(E, len(ShimEnv.STATS), generator=g) * 3
        stats0[:, ShimEnv.STATS.index("success")] = (stats0[:, 0] > 2).float()
        stats0[:, ShimEnv.STATS.index("first_capture_step")] = torch.where(stats0[:, 0] > 1.5, 300.0, 800.0)
        for i, k in enumerate(ShimEnv.STATS):
            env.stats[k] = stats0[:, i:i + 1].clone()
        td = env._compute_state_and_obs()
        ob, st = td[("agents", "observation")], td[("agents", "state")]
        obs_arrs.update({
            f"{tag}_pos": s["pos"], f"{tag}_rot": s["rot"], f"{tag}_vel": s["vel"], f"{tag}_throttle": thr,
            f"{tag}_target_pos": s["tpos"], f"{tag}_cyl": s["cyl"], f"{tag}_progress": progress,
            f"{tag}_detect_radius": np.float32(env.drone_detect_radius),
            f"{tag}_state_self": ob["state_self"], f"{tag}_state_others": ob["state_others"],
            f"{tag}_cylinders": ob["cylinders"], f"{tag}_state_drones": st["state_drones"],
            f"{tag}_state_cylinders": st["cylinders"], f"{tag}_drone_state": env.info["drone_state"],
            f"{tag}_full_state": env.drone_states, f"{tag}_blocked": env.blocked,
            f"{tag}_broadcast_detect": env.broadcast_detect, f"{tag}_cylinders_mask": env.cylinders_mask,
            f"{tag}_knn_mask": env.k_nearest_cylinders_mask})
        # reward on the same post-physics state
        env.action_error_order1 = torch.rand(E, A, generator=g)
        env.drone.throttle_difference = torch.rand(E, A, generator=g) * 0.2
        out = env._compute_reward_and_done()
        rew_arrs.update({
            f"{tag}_pos": s["pos"], f"{tag}_rot": s["rot"], f"{tag}_vel": s["vel"], f"{tag}_throttle": thr,
            f"{tag}_target_pos": s["tpos"], f"{tag}_cyl": s["cyl"], f"{tag}_progress": progress,
            f"{tag}_detect_radius": np.float32(env.drone_detect_radius),
            f"{tag}_use_deployment": np.int32(env.use_deployment), f"{tag}_smoothness_coef": np.float32(env.smoothness_coef),
            f"{tag}_stats0": stats0, f"{tag}_aerr": env.action_error_order1, f"{tag}_thr_diff": env.drone.throttle_difference,
            f"{tag}_reward": out[("agents", "reward")], f"{tag}_done": out["done"], f"{tag}_stats1": _stats_arr(env),
            f"{tag}_capture": env.capture})
    save("g_obs", **obs_arrs)
    save("g_reward", **rew_arrs)


def gen_grid():
    g = torch.Generator().manual_seed(20240970)
    E = 64
    xy = (torch.rand(E, 4, 2, generator=g) * 2 - 1) * 0.95
    center_pos = torch.zeros(E, 1, 2)
    center_grid = torch.ones(E, 1, 2, dtype=torch.int) * 4
    cells = hns_fn["continuous_to_grid"](xy, 9, 0.2, center_pos, center_grid)
    back = hns_fn["grid_to_continuous"](cells, 0.8, 0.2, center_pos, center_grid)
    gm = hns_fn["set_outside_circle_to_one"](torch.zeros(1, 9, 9, dtype=torch.int))
    save("g_grid", xy=xy, cells=cells, back=back, disc=gm[0])


# ---- the build's integrator spec in torch fp32; NOT reference code ----------
def integrate_spec(pos, rot, vel, force_w, torque_b, tpos, tvel, P):
    """Semi-implicit Euler in PhysX order; fp32, explicit component arithmetic.

    force_w: world force [E,A,3] (thrust rotated + downwash); torque_b: body torque [E,A,3].
    Exact elementary functions here are torch.sin/cos; the oracle uses its own polynomial
    sincos, so agreement is to fp32 rounding, not bitwise.
    """
    dt = torch.tensor(P["dt"], dtype=torch.float32)
    m = torch.tensor(P["mass"], dtype=torch.float32)
    I = torch.tensor(P["inertia"], dtype=torch.float32)
    lin, ang = vel[..., :3], vel[..., 3:]
    acc = force_w / m
    acc[..., 2] = acc[..., 2] - torch.tensor(P["g"], dtype=torch.float32)
    lin = (lin + acc * dt) * torch.tensor(max(0.0, 1.0 - P["dt"] * P["lin_damp"]), dtype=torch.float32)
    sp = torch.sqrt(lin[..., 0] ** 2 + lin[..., 1] ** 2 + lin[..., 2] ** 2)
    scale = torch.where(sp > P["v_max"], P["v_max"] / sp, torch.ones_like(sp))
    lin = lin * scale.unsqueeze(-1)
    wb = ref_torch.quat_rotate_inverse(rot, ang)
    Iw = wb * I
    gyro = torch.cross(wb, Iw, dim=-1)
    wb = (wb + (torque_b - gyro) / I * dt) * torch.tensor(max(0.0, 1.0 - P["dt"] * P["ang_damp"]), dtype=torch.float32)
    wn = torch.sqrt(wb[..., 0] ** 2 + wb[..., 1] ** 2 + wb[..., 2] ** 2)
    scale = torch.where(wn > P["w_max"], P["w_max"] / wn, torch.ones_like(wn))
    wb = wb * scale.unsqueeze(-1)
    ang = ref_torch.quat_rotate(rot, wb)
    pos = pos + lin * dt
    if P["ground"]:
        below = pos[..., 2] < 0.0
        pos[..., 2] = torch.where(below, torch.zeros_like(pos[..., 2]), pos[..., 2])
        lin[..., 2] = torch.where(below & (lin[..., 2] < 0), torch.zeros_like(lin[..., 2]), lin[..., 2])
    wn = torch.sqrt(ang[..., 0] ** 2 + ang[..., 1] ** 2 + ang[..., 2] ** 2)
    half = wn * dt * 0.5
    s_over = torch.where(wn > 1e-8, torch.sin(half) / wn, 0.5 * dt.expand_as(wn))
    dq = torch.cat([torch.cos(half).unsqueeze(-1), ang * s_over.unsqueeze(-1)], dim=-1)
    w1, x1, y1, z1 = dq.unbind(-1)
    w2, x2, y2, z2 = rot.unbind(-1)
    rot = torch.stack([((w1 * w2 - x1 * x2) - y1 * y2) - z1 * z2,
                       ((w1 * x2 + x1 * w2) + y1 * z2) - z1 * y2,
                       ((w1 * y2 - x1 * z2) + y1 * w2) + z1 * x2,
                       ((w1 * z2 + x1 * y2) - y1 * x2) + z1 * w2], dim=-1)
    qn = torch.sqrt(((rot[..., 0] ** 2 + rot[..., 1] ** 2) + rot[..., 2] ** 2) + rot[..., 3] ** 2)
    rot = rot / qn.unsqueeze(-1)
    tpos = tpos + tvel * dt
    return pos, rot, torch.cat([lin, ang], -1), tpos


PHYS = dict(dt=DT, g=9.81, mass=CF["mass"], inertia=[CF["inertia"]["xx"], CF["inertia"]["yy"], CF["inertia"]["zz"]],
            lin_damp=0.2, ang_damp=0.2, v_max=TASK["v_drone"] * (1.0 - 1e-6), w_max=1000.0, ground=1)


def gen_episode(tag, E, A, C, T, seed, max_len, action_scale=0.5, task=None, n_active=None, keep=None):
    """Closed loop: every stage is reference code except the integrator (A5 spec above).
    `keep`: store only these per-step records (the 100-step free-running fixtures)."""
    g = torch.Generator().manual_seed(seed)
    env = ShimEnv(E, A, C, task, max_len=max_len)
    tr = ShimTransform()
    s = rand_scene(g, E, A, C, n_active=n_active, spread=0.5)
    s["vel"] = s["vel"] * 0.1
    s["pos"][..., 2] = 0.5 + 0.2 * torch.rand(E, A, generator=g)
    s["tpos"][..., 2] = 0.5 + 0.2 * torch.rand(E, 1, generator=g)
    d = env.drone
    hover = math.sqrt(CF["mass"] * 9.81 / (4 * float(d.rotor_module.KF.data[0, 0, 0])))
    d.rotor_module.throttle.data.fill_(hover)
    d.set_state(s["pos"], s["rot"], s["vel"])
    env.target.pos = s["tpos"].clone()
    env.cylinders.pos = s["cyl"]
    env.progress_buf = torch.randint(0, 5, (E,), generator=g).float() + (max_len - T + 3)
    env.info["prev_action"][..., 3] = 0.575
    td0 = env._compute_state_and_obs()    # the reset-time obs pass (isaac_env.py:221)
    rec = {k: [] for k in ["action", "pos", "rot", "vel", "tpos", "tvel", "throttle", "integ", "last", "prev_action",
                           "progress", "stats", "aerr", "cmds", "force_w", "torque_b", "thr_diff",
                           "state_self", "state_others", "cylinders", "state_drones", "reward", "done"]}
    init = dict(pos=d.w_pos.clone(), rot=d.w_rot.clone(), vel=d.w_vel.clone(), tpos=env.target.pos.clone(),
                throttle=d.throttle.data.clone(), prev_action=env.info["prev_action"].clone(),
                progress=env.progress_buf.clone(), stats=_stats_arr(env).clone(), cyl=s["cyl"],
                state_self=td0[("agents", "observation")]["state_self"].clone())
    done = torch.zeros(E, 1, dtype=torch.bool)
    for t in range(T):
        action = torch.randn(E, A, 4, generator=g) * action_scale
        action[..., 3] += 0.3
        td = TensorDict({"agents": {"action": action.clone()},
                         "info": {"drone_state": env.info["drone_state"].clone(),
                                  "prev_action": env.info["prev_action"].clone()},
                         "stats": {}, "done": torch.zeros(E, 1, dtype=torch.bool)}, [E])
        td = tr.inv(td)                                   # A1 + A2
        cmds = td[("agents", "action")].clone()
        env._pre_sim_step(td)                             # A10, A3, A4, A6
        rotor_f = d.rotor_rec.calls[-1]["forces"].reshape(E, A, 4, 3)
        base = d.base_link.calls[-1]
        T_i = rotor_f[..., 2]
        ang = torch.tensor(CF["rotor_configuration"]["rotor_angles"], dtype=torch.float32)
        l = torch.tensor(CF["rotor_configuration"]["arm_lengths"], dtype=torch.float32)
        tsum = ((T_i[..., 0] + T_i[..., 1]) + T_i[..., 2]) + T_i[..., 3]
        tvec = torch.zeros(E, A, 3); tvec[..., 2] = tsum
        force_w = ref_torch.quat_rotate(d.w_rot, tvec) + base["forces"].reshape(E, A, 3)
        yaw = ref_torch.quat_rotate_inverse(d.w_rot, base["torques"].reshape(E, A, 3))[..., 2]
        sx, cx = torch.sin(ang) * l, torch.cos(ang) * l
        torque_b = torch.stack([
            ((sx[0] * T_i[..., 0] + sx[1] * T_i[..., 1]) + sx[2] * T_i[..., 2]) + sx[3] * T_i[..., 3],
            -(((cx[0] * T_i[..., 0] + cx[1] * T_i[..., 1]) + cx[2] * T_i[..., 2]) + cx[3] * T_i[..., 3]),
            yaw], dim=-1)
        tvel = env.target.vel[..., :3].clone()
        pos, rot, vel, tpos = integrate_spec(d.w_pos, d.w_rot, d.w_vel, force_w, torque_b, env.target.pos, tvel, PHYS)
        d.set_state(pos, rot, vel)                        # A5 (ours)
        env.target.pos = tpos
        env.progress_buf = env.progress_buf + 1
        tdo = env._compute_state_and_obs()                # A8
        out = env._compute_reward_and_done()              # A9
        ob = tdo[("agents", "observation")]
        for k, v in dict(action=action, pos=pos, rot=rot, vel=vel, tpos=tpos, tvel=tvel, throttle=d.throttle.data,
                         integ=tr.controller.integ.reshape(E, A, 3), last=tr.controller.last_body_rate.reshape(E, A, 3),
                         prev_action=env.info["prev_action"], progress=env.progress_buf, stats=_stats_arr(env),
                         aerr=env.action_error_order1, cmds=cmds, force_w=force_w, torque_b=torque_b,
                         thr_diff=d.throttle_difference, state_self=ob["state_self"], state_others=ob.get("state_others", torch.zeros(E, A, 0, 3)),
                         cylinders=ob["cylinders"], state_drones=tdo[("agents", "state")]["state_drones"],
                         reward=out[("agents", "reward")], done=out["done"]).items():
            rec[k].append(v.clone())
    save(f"g_episode_{tag}", **{"init_" + k: v for k, v in init.items()},
         **{k: torch.stack(v) for k, v in rec.items() if keep is None or k in keep},
         meta=np.array([E, A, C, T, max_len], dtype=np.int64))


def gen_episode_resetpid(tag, E, A, C, T, seed, max_len, reset_at, action_scale=0.5):
    """A closed-loop episode whose controller sees the REAL root `done` (transforms.py:449-454: reset_pid = tensordict['done']),
    crossing resets — VERDICT r3 #4.  As gen_episode, every stage of a step is reference code except the integrator; here the stepped
    tensordict carries `done` = the `done` the previous step returned (what torchrl's step_mdp hands back as the root), so envs whose
    episode ended keep pulsing reset_pid until they are reset.  At the steps in `reset_at` the done envs are reset as
    HideAndSeek._reset_idx (hideandseek.py:609-723) + MultirotorBase._reset_idx (multirotor.py:635-650) leave them — restated here,
    since those methods are Isaac view calls: placement from a task vector (drawn here, as the envgen path hands one to
    hns_reset_tasks), identity attitude (rpy box collapsed to zero), zero velocity, hover throttle, zeroed stats, first_capture_step
    set for ALL envs (:712), prev_action[..., 3] (:714-716), progress 0, root done False, the controller NOT touched, then the one
    extra physics step of the whole scene (:722-723: integrator spec with zero force and torque, the evader moving with the velocity it
    holds) and the observation pass (isaac_env.py:221)."""
    g = torch.Generator().manual_seed(seed)
    env = ShimEnv(E, A, C, None, max_len=max_len)
    tr = ShimTransform()
    s = rand_scene(g, E, A, C, spread=0.5)
    s["vel"] = s["vel"] * 0.1
    s["pos"][..., 2] = 0.5 + 0.2 * torch.rand(E, A, generator=g)
    s["tpos"][..., 2] = 0.5 + 0.2 * torch.rand(E, 1, generator=g)
    d = env.drone
    hover = math.sqrt(CF["mass"] * 9.81 / (4 * float(d.rotor_module.KF.data[0, 0, 0])))
    d.rotor_module.throttle.data.fill_(hover)
    d.set_state(s["pos"], s["rot"], s["vel"])
    env.target.pos = s["tpos"].clone()
    env.cylinders.pos = s["cyl"].clone()
    env.progress_buf = torch.randint(0, 4, (E,), generator=g).float() + (max_len - 6)     # episodes end at steps 2 .. 5
    env.info["prev_action"][..., 3] = 0.575
    env._compute_state_and_obs()
    names = ["action", "root_done", "cmds", "pos", "rot", "vel", "tpos", "tvel", "throttle", "integ", "last", "prev_action", "progress", "stats", "aerr",
             "state_self", "state_others", "cylinders", "state_drones", "reward", "done"]
    rec = {k: [] for k in names}
    init = dict(pos=d.w_pos.clone(), rot=d.w_rot.clone(), vel=d.w_vel.clone(), tpos=env.target.pos.clone(), throttle=d.throttle.data.clone(),
                prev_action=env.info["prev_action"].clone(), progress=env.progress_buf.clone(), stats=_stats_arr(env).clone(), cyl=s["cyl"].clone())
    resets = {k: [] for k in ["step", "mask", "tasks", "pos", "rot", "vel", "tpos", "cyl", "throttle", "prev_action", "progress", "stats", "integ", "last",
                              "state_self", "state_others", "cylinders", "state_drones"]}
    root_done = torch.zeros(E, 1, dtype=torch.bool)
    zero_f = torch.zeros(E, A, 3)
    for t in range(T):
        if t in reset_at:
            ids = root_done[:, 0].nonzero().squeeze(-1)
            assert 0 < len(ids) < E or t != reset_at[0], "the first reset must find some envs done and some still running"
            s2 = rand_scene(g, E, A, C, spread=0.5)
            s2["pos"][..., 2] = 0.5 + 0.2 * torch.rand(E, A, generator=g)
            s2["tpos"][..., 2] = 0.5 + 0.2 * torch.rand(E, 1, generator=g)
            tasks = torch.cat([s2["pos"].reshape(E, -1), s2["tpos"].reshape(E, -1), s2["cyl"].reshape(E, -1)], dim=1)
            pos, rot, vel = d.w_pos.clone(), d.w_rot.clone(), d.w_vel.clone()
            pos[ids] = s2["pos"][ids]
            rot[ids] = torch.tensor([1.0, 0.0, 0.0, 0.0])
            vel[ids] = 0.0
            env.target.pos[ids] = s2["tpos"][ids]
            cyl = env.cylinders.pos.clone()
            cyl[ids] = s2["cyl"][ids]
            env.cylinders.pos = cyl
            d.rotor_module.throttle.data[ids] = hover
            d.throttle_difference[ids] = 0.0
            for k in ShimEnv.STATS:
                env.stats[k][ids] = 0.0
            env.stats["first_capture_step"] = torch.ones_like(env.stats["first_capture_step"]) * max_len
            cmd_init = 2.0 * d.throttle.data[ids] ** 2 - 1.0
            env.info["prev_action"][ids, :, 3] = (0.5 * (CF["max_thrust_ratio"] + cmd_init)).mean(dim=-1)
            env.prev_actions[ids] = env.info["prev_action"][ids]
            env.progress_buf[ids] = 0.0
            # the extra sim.step(): the whole scene, no rotor forces
            pos, rot, vel, tpos = integrate_spec(pos, rot, vel, zero_f.clone(), zero_f.clone(), env.target.pos, env.target.vel[..., :3].clone(), PHYS)
            d.set_state(pos, rot, vel)
            env.target.pos = tpos
            tdo = env._compute_state_and_obs()
            root_done = root_done.clone()
            root_done[ids] = False
            ob = tdo[("agents", "observation")]
            mask = torch.zeros(E, dtype=torch.uint8)
            mask[ids] = 1
            for k, v in dict(step=torch.tensor(t), mask=mask, tasks=tasks, pos=pos, rot=rot, vel=vel, tpos=tpos, cyl=cyl, throttle=d.throttle.data,
                             prev_action=env.info["prev_action"], progress=env.progress_buf, stats=_stats_arr(env),
                             integ=tr.controller.integ.reshape(E, A, 3), last=tr.controller.last_body_rate.reshape(E, A, 3),
                             state_self=ob["state_self"], state_others=ob["state_others"], cylinders=ob["cylinders"],
                             state_drones=tdo[("agents", "state")]["state_drones"]).items():
                resets[k].append(v.clone())
        action = torch.randn(E, A, 4, generator=g) * action_scale
        action[..., 3] += 0.3
        td = TensorDict({"agents": {"action": action.clone()},
                         "info": {"drone_state": env.info["drone_state"].clone(), "prev_action": env.info["prev_action"].clone()},
                         "stats": {}, "done": root_done.clone()}, [E])
        td = tr.inv(td)                                   # A1 + A2 with reset_pid = the root done
        cmds = td[("agents", "action")].clone()          # what the transform hands to the env (transforms.py:455-456): the input of `action_input: motor`
        env._pre_sim_step(td)
        rotor_f = d.rotor_rec.calls[-1]["forces"].reshape(E, A, 4, 3)
        base = d.base_link.calls[-1]
        T_i = rotor_f[..., 2]
        ang = torch.tensor(CF["rotor_configuration"]["rotor_angles"], dtype=torch.float32)
        l = torch.tensor(CF["rotor_configuration"]["arm_lengths"], dtype=torch.float32)
        tsum = ((T_i[..., 0] + T_i[..., 1]) + T_i[..., 2]) + T_i[..., 3]
        tvec = torch.zeros(E, A, 3); tvec[..., 2] = tsum
        force_w = ref_torch.quat_rotate(d.w_rot, tvec) + base["forces"].reshape(E, A, 3)
        yaw = ref_torch.quat_rotate_inverse(d.w_rot, base["torques"].reshape(E, A, 3))[..., 2]
        sx, cx = torch.sin(ang) * l, torch.cos(ang) * l
        torque_b = torch.stack([
            ((sx[0] * T_i[..., 0] + sx[1] * T_i[..., 1]) + sx[2] * T_i[..., 2]) + sx[3] * T_i[..., 3],
            -(((cx[0] * T_i[..., 0] + cx[1] * T_i[..., 1]) + cx[2] * T_i[..., 2]) + cx[3] * T_i[..., 3]),
            yaw], dim=-1)
        tvel = env.target.vel[..., :3].clone()
        pos, rot, vel, tpos = integrate_spec(d.w_pos, d.w_rot, d.w_vel, force_w, torque_b, env.target.pos, tvel, PHYS)
        d.set_state(pos, rot, vel)
        env.target.pos = tpos
        env.progress_buf = env.progress_buf + 1
        tdo = env._compute_state_and_obs()
        out = env._compute_reward_and_done()
        ob = tdo[("agents", "observation")]
        for k, v in dict(action=action, root_done=root_done, cmds=cmds, pos=pos, rot=rot, vel=vel, tpos=tpos, tvel=tvel, throttle=d.throttle.data,
                         integ=tr.controller.integ.reshape(E, A, 3), last=tr.controller.last_body_rate.reshape(E, A, 3),
                         prev_action=env.info["prev_action"], progress=env.progress_buf, stats=_stats_arr(env), aerr=env.action_error_order1,
                         state_self=ob["state_self"], state_others=ob["state_others"], cylinders=ob["cylinders"],
                         state_drones=tdo[("agents", "state")]["state_drones"], reward=out[("agents", "reward")], done=out["done"]).items():
            rec[k].append(v.clone())
        root_done = out["done"].clone()                   # step_mdp: next.done becomes the root done of the next stepped tensordict
    assert torch.stack(rec["root_done"]).any(), "no reset_pid pulse in the sequence"
    save(f"g_episode_{tag}", **{"init_" + k: v for k, v in init.items()}, **{k: torch.stack(v) for k, v in rec.items()},
         **{"reset_" + k: torch.stack(v) for k, v in resets.items()}, meta=np.array([E, A, C, T, max_len], dtype=np.int64))


if __name__ == "__main__":
    gen_utils()
    gen_rotor()
    gen_pid()
    gen_downwash()
    gen_apply()
    gen_blocked()
    gen_prey()
    gen_obs_reward()
    gen_grid()
    gen_episode("a3c8", E=24, A=3, C=8, T=40, seed=20241001, max_len=60)
    gen_episode("a3c5", E=24, A=3, C=5, T=30, seed=20241002, max_len=60, n_active=0)
    gen_episode("a6c16", E=12, A=6, C=16, T=20, seed=20241003, max_len=60)
    # S_1 .. S_100 free running: the same closed loop, 100 steps, only what a free-running comparison reads
    FREE = ("action", "pos", "rot", "vel", "tpos", "tvel", "progress", "reward", "done", "state_self")
    gen_episode("free_a3c8", E=16, A=3, C=8, T=100, seed=20250301, max_len=120, keep=FREE)
    gen_episode("free_a6c16", E=8, A=6, C=16, T=100, seed=20250302, max_len=120, keep=FREE)
    # the controller's reset_pid fed with the real root `done`, across two resets
    gen_episode_resetpid("resetpid_a3c5", E=16, A=3, C=5, T=14, seed=20260927, max_len=40, reset_at=(4, 9))


# --------------------------------------------------------------------------------------
# Hover (config 1): omni_drones/envs/single/hover.py
# --------------------------------------------------------------------------------------
HOVER = "omni_drones/envs/single/hover.py"
HOVER_TASK = yaml.safe_load(open(os.path.join(REF, "cfg/task/Hover.yaml")))
hov_ns = dict(torch=torch, TensorDict=TensorDict, TensorDictBase=TensorDict, quat_rotate_inverse=ref_torch.quat_rotate_inverse,
              collections=__import__("collections"))
hov_m = exec_functions(extract_source(HOVER, ["_pre_sim_step", "_compute_state_and_obs", "_compute_reward_and_done"],
                                      classname="Hover"), hov_ns)
HOVER_STATS = ["return", "pos_bonus", "head_bonus", "reward_pos", "reward_up", "reward_vel", "reward_acc", "reward_jerk",
               "episode_len", "pos_error", "heading_alignment", "uprightness", "action_smoothness", "linear_v_max",
               "angular_v_max", "linear_a_max", "angular_a_max", "linear_jerk_max", "angular_jerk_max", "linear_v_mean",
               "angular_v_mean", "linear_a_mean", "angular_a_mean", "linear_jerk_mean", "angular_jerk_mean", "motor1",
               "motor2", "motor3", "motor4", "cmd_r", "cmd_p", "cmd_y", "cmd_thrust", "target_r_rate", "target_p_rate",
               "target_y_rate", "real_r_rate", "real_p_rate", "real_y_rate"]
HOVER_ACC = ["linear_v_episode", "angular_v_episode", "linear_a_episode", "angular_a_episode", "linear_jerk_episode",
             "angular_jerk_episode", "last_linear_v", "last_angular_v", "last_linear_a", "last_angular_a",
             "last_linear_jerk", "last_angular_jerk"]


class ShimHover:
    """`self` for the Hover methods (attributes per hover.py:76-155)."""

    def __init__(self, E, max_len=500):
        t = HOVER_TASK
        self.num_envs, self.device, self.batch_size = E, "cpu", [E]
        self.max_episode_length = max_len
        self.dt = DT
        self.cfg = types.SimpleNamespace(task=types.SimpleNamespace(
            action_noise=False, omega=False, motor=False, add_noise=False))
        self.time_encoding, self.time_encoding_dim, self.latency = True, 4, 0
        self.reward_distance_scale = t["reward_distance_scale"]
        self.reward_v_scale, self.reward_acc_scale, self.reward_jerk_scale = t["reward_v_scale"], t["reward_acc_scale"], t["reward_jerk_scale"]
        self.linear_vel_max, self.linear_acc_max = t["linear_vel_max"], t["linear_acc_max"]
        self.alpha = 0.8
        self.drone = ShimDrone(E, 1)
        self.drone.intrinsics = torch.zeros(E, 1, 1)
        self.target_pos = torch.tensor([[0.0, 0.0, 1.0]])
        self.target_heading = torch.zeros(E, 1, 3)
        self.target_heading[..., 0] = 1.0
        self.progress_buf = torch.zeros(E)
        self.stats = TensorDict({k: torch.zeros(E, 1) for k in HOVER_STATS}, [E])
        self.info = TensorDict({"drone_state": torch.zeros(E, 1, 13), "prev_action": torch.zeros(E, 1, 4)}, [E])
        for k in HOVER_ACC:
            setattr(self, k, torch.zeros(E, 1))

    def acc(self):
        return torch.cat([getattr(self, k) for k in HOVER_ACC], dim=1)


def gen_hover():
    g = torch.Generator().manual_seed(20241010)
    E, T, max_len = 16, 30, 40
    env = ShimHover(E, max_len)
    tr = ShimTransform()
    d = env.drone
    pos = (torch.rand(E, 1, 3, generator=g) * 2 - 1) * torch.tensor([1.0, 1.0, 0.0]) + torch.tensor([0.0, 0.0, 0.05]) \
        + torch.rand(E, 1, 3, generator=g) * torch.tensor([0.0, 0.0, 1.95])
    pos[0, 0] = torch.tensor([0.005, -0.004, 1.003])     # inside the 0.02 bonus ball
    rot = rand_quat(g, E, 1, max_tilt=0.5)
    rot[0, 0] = torch.tensor([1.0, 0.0, 0.0, 0.0])
    vel = torch.randn(E, 1, 6, generator=g) * 0.2
    hover = math.sqrt(CF["mass"] * 9.81 / (4 * float(d.rotor_module.KF.data[0, 0, 0])))
    d.rotor_module.throttle.data.fill_(hover)
    d.set_state(pos, rot, vel)
    env.progress_buf = torch.randint(0, 4, (E,), generator=g).float() + (max_len - T + 5)
    env._compute_state = lambda: hov_m["_compute_state_and_obs"](env)
    td0 = hov_m["_compute_state_and_obs"](env)
    init = dict(pos=pos, rot=rot, vel=vel, throttle=d.throttle.data.clone(), progress=env.progress_buf.clone(),
                stats=torch.cat([env.stats[k] for k in HOVER_STATS], 1), acc=env.acc(),
                obs=td0[("agents", "observation")].clone())
    rec = {k: [] for k in ["action", "pos", "rot", "vel", "throttle", "integ", "last", "prev_action", "progress", "stats",
                           "acc", "obs", "reward", "done"]}
    prev = torch.zeros(E, 1, 4)
    for t in range(T):
        action = torch.randn(E, 1, 4, generator=g) * 0.5
        action[..., 3] += 0.3
        td = TensorDict({"agents": {"action": action.clone()},
                         "info": {"drone_state": env.info["drone_state"].clone(), "prev_action": prev.clone()},
                         "stats": {}, "done": torch.zeros(E, 1, dtype=torch.bool)}, [E])
        td = tr.inv(td)
        prev = td[("info", "prev_action")].clone()
        hov_m["_pre_sim_step"](env, td)
        rotor_f = d.rotor_rec.calls[-1]["forces"].reshape(E, 1, 4, 3)
        base = d.base_link.calls[-1]
        T_i = rotor_f[..., 2]
        ang = torch.tensor(CF["rotor_configuration"]["rotor_angles"], dtype=torch.float32)
        l = torch.tensor(CF["rotor_configuration"]["arm_lengths"], dtype=torch.float32)
        tsum = ((T_i[..., 0] + T_i[..., 1]) + T_i[..., 2]) + T_i[..., 3]
        tvec = torch.zeros(E, 1, 3); tvec[..., 2] = tsum
        force_w = ref_torch.quat_rotate(d.w_rot, tvec) + base["forces"].reshape(E, 1, 3)
        yaw = ref_torch.quat_rotate_inverse(d.w_rot, base["torques"].reshape(E, 1, 3))[..., 2]
        sx, cx = torch.sin(ang) * l, torch.cos(ang) * l
        torque_b = torch.stack([
            ((sx[0] * T_i[..., 0] + sx[1] * T_i[..., 1]) + sx[2] * T_i[..., 2]) + sx[3] * T_i[..., 3],
            -(((cx[0] * T_i[..., 0] + cx[1] * T_i[..., 1]) + cx[2] * T_i[..., 2]) + cx[3] * T_i[..., 3]),
            yaw], dim=-1)
        phys = dict(PHYS); phys["v_max"] = 1000.0          # Hover keeps the default max_linear_velocity (robots/config.py:36)
        npos, nrot, nvel, _ = integrate_spec(d.w_pos, d.w_rot, d.w_vel, force_w, torque_b, torch.zeros(E, 1, 3), torch.zeros(E, 1, 3), phys)
        d.set_state(npos, nrot, nvel)
        env.progress_buf = env.progress_buf + 1
        tdo = hov_m["_compute_state_and_obs"](env)
        out = hov_m["_compute_reward_and_done"](env)
        for k, v in dict(action=action, pos=npos, rot=nrot, vel=nvel, throttle=d.throttle.data,
                         integ=tr.controller.integ.reshape(E, 1, 3), last=tr.controller.last_body_rate.reshape(E, 1, 3),
                         prev_action=prev, progress=env.progress_buf, stats=torch.cat([env.stats[k] for k in HOVER_STATS], 1),
                         acc=env.acc(), obs=tdo[("agents", "observation")], reward=out[("agents", "reward")],
                         done=out["done"]).items():
            rec[k].append(v.clone())
    save("g_hover", **{"init_" + k: v for k, v in init.items()}, **{k: torch.stack(v) for k, v in rec.items()},
         meta=np.array([E, T, max_len], dtype=np.int64))


# --------------------------------------------------------------------------------------
# TP_net inside the observation: omni_drones/learning/mappo.py:572-589 +
# hideandseek.py:805-854.  TP_net is a plain nn.Module; its class is AST-extracted because the
# module top of mappo.py imports torchrl.
# --------------------------------------------------------------------------------------
tp_ns = dict(torch=torch, nn=torch.nn)
tp_cls = None


def _tp_class():
    global tp_cls
    if tp_cls is None:
        src = open(os.path.join(REF, "omni_drones/learning/mappo.py")).read()
        tree = ast.parse(src)
        node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TP_net"][0]
        exec(compile(ast.get_source_segment(src, node), "<ref:TP_net>", "exec"), tp_ns)
        tp_cls = tp_ns["TP_net"]
    return tp_cls


def gen_tp_obs(tag="g_tp_obs", E=48, A=3, C=5, seed=20241020, use_obstacles=0):
    import collections
    g = torch.Generator().manual_seed(seed)
    T = 14
    env = ShimEnv(E, A, C, {"drone_detect_radius": 0.9}, max_len=20)
    env.use_TP_net = 1
    env.use_obstacles = use_obstacles                 # hideandseek.py:808-816: the frame also holds the cylinders
    env.future_predcition_step, env.history_step, env.window_step = 5, 10, 1
    torch.manual_seed(123)
    env.TP = _tp_class()(input_dim=1 + 3 + 3 + 3 * A + (3 * C if use_obstacles else 0), output_dim=15, future_predcition_step=5, window_step=1)
    env.history_data = collections.deque(maxlen=10)
    weights = {k: v.detach().clone() for k, v in env.TP.state_dict().items()}
    rec = {k: [] for k in ["pos", "rot", "vel", "throttle", "tpos", "tvel", "progress", "state_self", "state_drones",
                           "TP_input", "TP_groundtruth", "TP_done", "broadcast_detect"]}
    cyl = rand_scene(g, E, A, C)["cyl"]
    env.cylinders.pos = cyl
    for t in range(T):
        s = rand_scene(g, E, A, C)
        thr = torch.rand(E, A, 4, generator=g)
        env.drone.rotor_module.throttle.data.copy_(thr)
        env.drone.set_state(s["pos"], s["rot"], s["vel"])
        env.target.pos = s["tpos"]
        env.target.vel = torch.randn(E, 1, 6, generator=g)
        env.progress_buf = torch.full((E,), float(t + 6))
        with torch.no_grad():
            td = env._compute_state_and_obs()
        tp = td[("agents", "TP")]
        for k, v in dict(pos=s["pos"], rot=s["rot"], vel=s["vel"], throttle=thr, tpos=s["tpos"], tvel=env.target.vel[..., :3],
                         progress=env.progress_buf, state_self=td[("agents", "observation")]["state_self"],
                         state_drones=td[("agents", "state")]["state_drones"], TP_input=tp["TP_input"],
                         TP_groundtruth=tp["TP_groundtruth"], TP_done=tp["TP_done"], broadcast_detect=env.broadcast_detect).items():
            rec[k].append(v.clone())
    save(tag, cyl=cyl, **{k: torch.stack(v) for k, v in rec.items()},
         **{"w_" + k.replace(".", "_"): v for k, v in weights.items()}, meta=np.array([E, A, C, T, 20], dtype=np.int64),
         use_obstacles=np.array(use_obstacles, dtype=np.int64))


if __name__ == "__main__":
    gen_hover()
    gen_tp_obs()
    gen_tp_obs("g_tp_obs_a6", E=10, A=6, C=8, seed=20241021)      # 25-value frames: the two-chunk path of the HIP kernel
    gen_tp_obs("g_tp_obs_obst", E=12, A=3, C=5, seed=20241023, use_obstacles=1)   # task.use_obstacles: 31-value frames
    gen_tp_obs("g_tp_obs_obst_c8", E=12, A=3, C=8, seed=20241027, use_obstacles=1)   # 40-value frames: the three-chunk path


# ---- envgen grid sanity check (hideandseek_envgen.py:145-207), module-level functions executed as they are ----
def gen_envgen_sanity():
    ENVGEN = "omni_drones/envs/hide_and_seek/hideandseek_envgen.py"
    ns = dict(torch=torch, np=np)
    fn = exec_functions(extract_source(ENVGEN, ["continuous_to_grid", "set_outside_circle_to_one", "sanity_check"]), ns)
    rng = np.random.default_rng(20241022)
    A, C, N = 3, 5, 3000
    grid_size, num_grid = 0.2, 9
    grid_map = fn["set_outside_circle_to_one"](np.zeros((1, num_grid, num_grid), dtype=int))
    center_pos, center_grid = np.zeros((1, 2)), np.ones((1, 2), dtype=int) * (num_grid // 2)
    tasks = np.zeros((N, 3 * (A + 1 + C)), dtype=np.float32)
    xy = rng.uniform(-0.75, 0.75, size=(N, A + 1 + C, 2))
    snap = rng.random((N, A + 1 + C, 1)) < 0.5                     # half of the bodies sit exactly on cell centres / edges
    xy = np.where(snap, np.round(xy / 0.1) * 0.1, xy)
    tasks.reshape(N, -1, 3)[..., :2] = xy
    tasks.reshape(N, -1, 3)[..., 2] = rng.uniform(-20, 1.3, size=(N, A + 1 + C))
    tasks[: N // 3, 3 * (A + 1):] = np.tile(np.array([[0.0, 0.0, -20.0], [0.2, 0.0, 0.6], [0.0, 0.2, 0.6], [-0.2, 0.0, 0.6], [0.0, -0.2, 0.6]],
                                                   np.float32).reshape(-1), (N // 3, 1))   # a sane cylinder set: the bodies decide
    ok = np.zeros(N, dtype=bool)
    t64 = tasks.astype(np.float64)
    for i in range(N):
        d = t64[i, :3 * A].reshape(-1, 3)
        t = t64[i, 3 * A:3 * A + 3].reshape(-1, 3)
        c = t64[i, 3 * A + 3:].reshape(-1, 3)
        g = [fn["continuous_to_grid"](torch.from_numpy(x[..., :2]), num_grid, grid_size, torch.from_numpy(center_pos), torch.from_numpy(center_grid)).numpy()
             for x in (d, t, c)]
        ok[i] = bool(fn["sanity_check"](grid_map[0], *g))
    save("g_envgen_sanity", tasks=tasks, ok=ok, disc=grid_map[0], meta=np.array([A, C, N], dtype=np.int64))


if __name__ == "__main__":
    gen_envgen_sanity()


# ---- key / shape / dtype manifest of the env boundary ------------------------------------------
# The reference's own `_set_specs` (hideandseek.py:327-433) is executed against recording spec shims, and the
# reference's own `IsaacEnv._reset` / `_step` (isaac_env.py:210-240) around its `_compute_state_and_obs` /
# `_compute_reward_and_done` against the tensor shims above: what comes out is the key tree a caller of
# `env.reset()` / `env.step()` sees, with shapes and dtypes, for use_TP_net 0 and 1.  Only names/shapes/dtypes are stored.
class _Spec:
    def __init__(self, shape, dtype="float32", kind="unbounded", low=None, high=None):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        self.shape, self.dtype, self.kind, self.low, self.high = torch.Size(shape), dtype, kind, low, high

    def expand(self, *sizes):
        sizes = sizes[0] if len(sizes) == 1 and isinstance(sizes[0], (tuple, list, torch.Size)) else sizes
        return _Spec((*sizes, *self.shape), self.dtype, self.kind, self.low, self.high)

    def to(self, device):
        return self

    def zero(self):
        return torch.zeros(self.shape)

    @classmethod
    def __torch_function__(cls, func, types_, args=(), kwargs=None):
        if func is torch.stack:                      # torch.stack([spec] * n, dim=0)  (hideandseek.py:382,428)
            specs = args[0]
            return _Spec((len(specs), *specs[0].shape), specs[0].dtype, specs[0].kind, specs[0].low, specs[0].high)
        return NotImplemented


class _Composite(dict):
    def __init__(self, d=None, shape=()):
        super().__init__()
        self.shape = torch.Size(shape)
        for k, v in (d or {}).items():
            self[k] = v

    def expand(self, *sizes):
        sizes = sizes[0] if len(sizes) == 1 and isinstance(sizes[0], (tuple, list, torch.Size)) else sizes
        return _Composite({k: v.expand(*sizes) for k, v in self.items()}, (*sizes, *self.shape))

    def to(self, device):
        return self

    def zero(self):
        return TensorDict({k: v.zero() for k, v in self.items()}, list(self.shape))


def _spec_tree(spec):
    if isinstance(spec, _Composite):
        return {k: _spec_tree(v) for k, v in spec.items()}
    out = {"shape": list(spec.shape), "dtype": spec.dtype, "kind": spec.kind}
    if spec.low is not None:
        out["low"], out["high"] = spec.low, spec.high
    return out


def _td_tree(td):
    out = {}
    for k, v in td.items():
        if isinstance(v, dict):
            out[k] = _td_tree(v)
        else:
            out[k] = {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", "")}
    return out


def gen_manifest(E=6, A=3, C=5):
    import collections
    import json
    ns = {"torch": torch, "CompositeSpec": _Composite,
          "UnboundedContinuousTensorSpec": lambda shape, device=None: _Spec(shape),
          "AgentSpec": lambda name, n, **keys: {"name": name, "n": n, **{k: list(v) for k, v in keys.items()}}}
    set_specs = exec_functions(extract_source("omni_drones/envs/hide_and_seek/hideandseek.py", ["_set_specs"], "HideAndSeek"), ns)["_set_specs"]
    base = exec_functions(extract_source("omni_drones/envs/isaac_env.py", ["_reset", "_step"], "IsaacEnv"),
                          {"torch": torch, "TensorDict": TensorDict, "TensorDictBase": TensorDict, "Optional": None})
    manifest = {"meta": {"num_envs": E, "num_agents": A, "num_cylinders": C, "obs_max_cylinder": 3, "history_step": 10, "future_predcition_step": 5,
                         "source": "hideandseek.py:327-433 (_set_specs), isaac_env.py:210-240 (_reset/_step), :746-917, :919-1065 executed on shims"}}
    for tp in (0, 1):
        env = ShimEnv(E, A, C, {"drone_detect_radius": 0.9}, max_len=20)
        env.use_TP_net = tp
        # what _set_specs reads (hideandseek.py:327-335; drone specs: multirotor.py state 23 values, action 4 values in [-1, 1])
        env.drone.state_spec = _Spec(23)
        env.drone.action_spec = _Spec(4, kind="bounded", low=-1.0, high=1.0)
        env.drone.n = A
        t = env.cfg.task
        t.time_encoding, t.future_predcition_step, t.history_step, t.window_step, t.use_obstacles = True, 5, 10, 1, 0
        t.cylinder = types.SimpleNamespace(obs_max_cylinder=3)
        env.agent_spec = {}
        set_specs(env)
        specs = {"observation_spec": _spec_tree(env.observation_spec), "action_spec": _spec_tree(env.action_spec),
                 "reward_spec": _spec_tree(env.reward_spec), "agent_spec": env.agent_spec["drone"]}
        # runtime trees: the reference's own _reset/_step wrappers around its obs / reward passes
        env2 = ShimEnv(E, A, C, {"drone_detect_radius": 0.9}, max_len=20)
        env2.use_TP_net = tp
        env2.future_predcition_step, env2.history_step, env2.window_step = 5, 10, 1
        if tp:
            torch.manual_seed(1)
            env2.TP = _tp_class()(input_dim=1 + 3 + 3 + 3 * A, output_dim=15, future_predcition_step=5, window_step=1)
            env2.history_data = collections.deque(maxlen=10)
        g = torch.Generator().manual_seed(5)
        s = rand_scene(g, E, A, C)
        env2.cylinders.pos = s["cyl"]
        env2.drone.set_state(s["pos"], s["rot"], s["vel"])
        env2.target.pos = s["tpos"]
        env2.target.vel = torch.zeros(E, 1, 6)
        env2._reset_idx = lambda env_ids: None
        env2.sim = types.SimpleNamespace(_physics_sim_view=types.SimpleNamespace(flush=lambda: None), step=lambda render=False: None)
        env2._post_sim_step = lambda td: None
        env2._pre_sim_step = lambda td: None
        env2.action_error_order1 = torch.zeros(E, A)       # what _pre_sim_step leaves behind (:731)
        with torch.no_grad():
            reset_td = base["_reset"](env2, None)
            step_td = base["_step"](env2, TensorDict({}, [E]))
        manifest[f"use_TP_net={tp}"] = {"specs": specs, "reset": _td_tree(reset_td), "step": _td_tree(step_td)}
    with open(os.path.join(OUT, "g_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote g_manifest.json")


if __name__ == "__main__":
    gen_manifest()


# ---- A12: the generator's buffer and the curriculum block, executed as the reference wrote them ----------------------------------
# `GenBuffer` (hideandseek_envgen.py:209-377) is extracted as a whole class; the curriculum statements of `_compute_reward_and_done`
# (:1241-1246 success_buffer / success_unif, :1302-1333 `if torch.any(done): ...`, :1335-1336) are extracted as AST statements and exec'd against a
# namespace `self`.  DGL is not installed (and unpinned in the reference): `farthest_point_sampler` is an exact FPS whose start index the
# generator names (DGL draws it at random) — ties to the lower index; the fixture's point sets are checked to give the same selection in fp32 and fp64.
def _stmt_sources(relpath, classname, method, wanted):
    """Source text of the top-level statements of `classname.method` for which `wanted(src_of_statement)` is true, in order."""
    src = open(os.path.join(REF, relpath)).read()
    tree = ast.parse(src)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == classname][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == method][0]
    import textwrap
    out = []
    for node in fn.body:
        seg = ast.get_source_segment(src, node)
        if wanted(seg):
            out.append(textwrap.dedent(" " * node.col_offset + seg))
    return out


def _exact_fps(points, k, start):
    n = points.shape[0]
    idx = np.empty(k, dtype=np.int64)
    dist = np.full(n, np.inf, dtype=points.dtype)
    cur = int(start)
    for i in range(k):
        idx[i] = cur
        d = ((points - points[cur]) ** 2).sum(-1)
        dist = np.minimum(dist, d)
        dist[cur] = -1.0
        cur = int(np.argmax(dist))
    return idx


def gen_genbuffer():
    import collections
    import copy
    ENVGEN = "omni_drones/envs/hide_and_seek/hideandseek_envgen.py"
    src = open(os.path.join(REF, ENVGEN)).read()
    tree = ast.parse(src)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GenBuffer"][0]
    fps_log = {}

    def farthest_point_sampler(pos, npoints, start_idx=None):
        pts = pos[0].numpy()
        a = _exact_fps(pts, npoints, fps_log["start"])
        b = _exact_fps(pts.astype(np.float64), npoints, fps_log["start"])
        assert np.array_equal(a, b), "fixture point set has a near-tie: fp32 and fp64 FPS disagree; change the seed"
        fps_log["normed"] = pts.copy()
        fps_log["idx"] = a
        return torch.from_numpy(a)[None]

    ns = dict(torch=torch, np=np, math=math, copy=copy, deque=collections.deque, farthest_point_sampler=farthest_point_sampler)
    exec_functions(extract_source(ENVGEN, ["select_unoccupied_positions", "grid_to_continuous", "continuous_to_grid",
                                           "set_outside_circle_to_one", "sanity_check"]), ns)
    # `init_easy_cases` hands its numpy grid map to `select_unoccupied_positions`, whose `torch.nonzero(...)` takes tensors only (:112): as written the
    # reference's `use_init_easy: 1` raises a TypeError.  The one adaptation made here: the map is converted on the way in; everything else is as written.
    # Likewise it hands numpy `center_pos` / `center_grid` to `grid_to_continuous`, which adds them to a tensor and clamps with torch (:136-141): converted too.
    # (`use_init_easy` is 0 in the reference's task file; the method is pinned here as far as it can be made to run.)
    _select, _g2c = ns["select_unoccupied_positions"], ns["grid_to_continuous"]
    ns["select_unoccupied_positions"] = lambda occ, n: _select(torch.as_tensor(occ), n)
    ns["grid_to_continuous"] = lambda g, b, gs, cp, cg: _g2c(g, b, gs, torch.as_tensor(cp), torch.as_tensor(cg))
    exec(compile(ast.get_source_segment(src, cls), "<ref:GenBuffer>", "exec"), ns)
    GenBuffer = ns["GenBuffer"]
    out = {}

    # (1) init_easy_cases: the start cell the reference drew and the cells its flood gave the pursuers
    # The flood appends EVERY free neighbour of the cell it expands and stops only at `len(found) == 4`, so with fewer than four pursuers `found` holds
    # up to four cells (ragged rows / a shape mismatch with the z column): as written the method runs for num_agents == 4 only.  `easy_runs` records that.
    runs = []
    for A in (1, 2, 3, 4, 5):
        torch.manual_seed(20241100 + A)
        gb = GenBuffer(A, 5, "cpu")
        gb.buffer_length = 160
        try:
            easy = gb.init_easy_cases().numpy()                   # [160, A + 1, 3]: pursuers..., evader
        except (ValueError, RuntimeError, TypeError):
            runs.append(0)
            continue
        runs.append(1)
        out[f"easy_a{A}"] = easy.astype(np.float64)
        assert easy.shape == (160, A + 1, 3)
    out["easy_runs"] = np.array(runs, dtype=np.int64)             # for A = 1..5
    assert runs[3] == 1
    out["easy_disc"] = gb.grid_map[0].copy()

    # (2) the bounds `samplenearby` clips to (:320-333), by executing its own statements up to `boundary_task = np.array(boundary_task)`
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "samplenearby"][0]
    for A, Cn in ((3, 5), (4, 5)):
        gb = GenBuffer(A, Cn, "cpu")
        env_ns = dict(ns, self=gb)
        import textwrap
        for node in fn.body:
            seg = ast.get_source_segment(src, node)
            if isinstance(node, ast.Assign) and ("boundary" in seg) or (isinstance(node, ast.AugAssign) and "boundary_task" in seg):
                exec(textwrap.dedent(" " * node.col_offset + seg), env_ns)
        out[f"bounds_a{A}c{Cn}"] = np.asarray(env_ns["boundary_task"], dtype=np.float64)

    # (3) samplenearby itself: outputs of the reference on a seeded history (RNG stream unpinned: the test checks properties of THESE outputs)
    A, Cn = 3, 5
    gb = GenBuffer(A, Cn, "cpu")
    rng = np.random.default_rng(20241101)

    def valid_tasks(n):
        tasks = []
        free = np.argwhere(gb.grid_map[0] == 0)
        while len(tasks) < n:
            cells = free[rng.permutation(len(free))[:A + 1 + Cn]]
            xy = (cells - gb.num_grid // 2) * gb.grid_size + rng.uniform(-0.04, 0.04, size=(A + 1 + Cn, 2))
            z = np.concatenate([rng.uniform(1.1, 1.3, size=A + 1), np.where(rng.random(Cn) < 0.7, 0.6, -20.0)])
            tasks.append(np.concatenate([xy, z[:, None]], axis=1).reshape(-1))
        return np.asarray(tasks, dtype=np.float32)

    hist = valid_tasks(64)
    gb._history_buffer = hist.copy()
    np.random.seed(20241102)
    for expand in (0, 1):
        near = gb.samplenearby(200, expand, 0.1)
        assert near.shape == (200, gb.task_dim)
        out[f"near_expand{expand}"] = near.astype(np.float64)
    out["near_history"] = hist

    # (4) two task batches through insert -> insert_weights x eval_iter -> the curriculum block (update, statistics, R_min..R_max filter, insert_history)
    E, eval_iter, R_min, R_max = 96, 3, 0.3, 0.7
    blocks = _stmt_sources(ENVGEN, "HideAndSeek_envgen", "_compute_reward_and_done",
                           lambda s: s.startswith("if self.num_unif < self.num_envs") or s.startswith("if torch.any(done)")
                           or s.startswith('self.stats["history_buffer"]') or s.startswith('self.stats["ratio_unif"]'))
    assert len(blocks) == 4, [b[:40] for b in blocks]
    gb = GenBuffer(A, Cn, "cpu")
    gb.buffer_length = 80                                         # so that the second batch's insert_history trims by FPS
    names = ["success", "success_buffer", "success_unif", "history_buffer", "add_history", "ratio_unif"]
    names += [f"ratio_cylinders_{i}" for i in range(Cn + 1)] + [f"success_cylinders_{i}" for i in range(Cn + 1)]
    self = types.SimpleNamespace(gen_buffer=gb, stats={k: torch.zeros(E, 1) for k in names}, num_envs=E, num_unif=E, num_cylinders=Cn,
                                 ratio_unif=0.3, success_threshold=1.0, update_iter=0, eval_iter=eval_iter, R_min=R_min, R_max=R_max,
                                 device="cpu", active_cylinders=None)
    env_ns = dict(ns, self=self, done=torch.ones(E, 1, dtype=torch.bool))
    for batch in range(2):
        tasks = valid_tasks(E)
        self.num_unif = E if batch == 0 else 40
        self.active_cylinders = torch.from_numpy((tasks.reshape(E, -1, 3)[:, A + 1:, 2] > 0.0).sum(-1, keepdims=True).astype(np.float32))
        gb.insert(tasks)                                           # :895
        out[f"b{batch}_tasks"] = tasks
        out[f"b{batch}_num_unif"] = np.int64(self.num_unif)
        out[f"b{batch}_active"] = self.active_cylinders.numpy().copy()
        p_env = rng.uniform(0.05, 0.95, size=E)                    # every env's own success probability: weights spread over [0, 1]
        fps_log["start"] = 17 + batch
        for ep in range(eval_iter):
            self.stats["success"] = torch.from_numpy((rng.random(E) < p_env).astype(np.float32)).unsqueeze(1)
            out[f"b{batch}_success{ep}"] = self.stats["success"].numpy().copy()
            fps_log.pop("idx", None)
            for code in blocks:
                exec(code, env_ns)
            for k in names[1:]:
                v = self.stats[k].numpy().astype(np.float64)
                assert v.shape == (E, 1)
                out[f"b{batch}_ep{ep}_{k}"] = v[0, 0].copy() if (v == v[0, 0]).all() else v.copy()   # [E, 1] filled with one value -> that value
            out[f"b{batch}_ep{ep}_update_iter"] = np.int64(self.update_iter)
        out[f"b{batch}_weight_buffer"] = np.asarray(gb._weight_buffer, dtype=np.float64).copy()
        out[f"b{batch}_state_buffer"] = np.asarray(gb._state_buffer).copy()
        out[f"b{batch}_history"] = np.asarray(gb._history_buffer).copy()
        out[f"b{batch}_fps_start"] = np.int64(fps_log["start"] if "idx" in fps_log else -1)
        if "idx" in fps_log:
            out[f"b{batch}_fps_idx"] = fps_log["idx"].copy()
            out[f"b{batch}_fps_normed"] = fps_log["normed"].copy()
    assert "b1_fps_idx" in out and out["b0_fps_start"] == -1 and 0 < out["b0_ep2_add_history"] < E
    out["meta"] = np.array([A, Cn, E, eval_iter, 80], dtype=np.int64)
    out["r_bounds"] = np.array([R_min, R_max])
    # (5) success above success_threshold switches the generator to uniform tasks only (:1303-1304)
    self.success_threshold = 0.5
    self.stats["success"] = torch.ones(E, 1)
    gb.insert(valid_tasks(E))
    for code in blocks:
        exec(code, env_ns)
    out["ratio_unif_after_threshold"] = np.float64(self.ratio_unif)
    save("g_genbuffer", **out)


def gen_learner_moments(E=64, T=8, A=3, rollouts=3, seed=20261001):
    """The consumer of the path's ONE collective, executed as the reference wrote it: the advantage-normalisation statements of MAPPOPolicy.train_op
    (learning/mappo.py:391-396: `(adv - mean) / (std + 1e-8)`, torch.std = unbiased) and the ValueNorm1 update / normalize that follow them (:398-402 ->
    learning/utils/valuenorm.py:83-98, loaded by path; cfg/algo/mappo.yaml:44-49: critic.value_norm = ValueNorm1, beta 0.995, input_shape = reward_spec.shape[-1:] = (1,), mappo.py:215-218).
    The statements run on whole `[E, T, A, 1]` tensors; tests split the same tensors over 1 / 2 / 8 ranks and must land on these outputs."""
    MAPPO = "omni_drones/learning/mappo.py"
    blocks = _stmt_sources(MAPPO, "MAPPOPolicy", "train_op",
                           lambda s: s.startswith("advantages_mean") or s.startswith("advantages_std") or s.startswith("if self.normalize_advantages")
                           or (s.startswith("if hasattr(self, \"value_normalizer\")") and "value_normalizer.update" in s))
    assert len(blocks) == 4, blocks
    valuenorm = load_by_path("ref_valuenorm", "omni_drones/learning/utils/valuenorm.py")
    algo = yaml.safe_load(open(os.path.join(REF, "cfg/algo/mappo.yaml")))
    vcfg = algo["critic"]["value_norm"]                  # (mappo.py:158 hands cfg.critic to the critic's constructor, which reads value_norm at :208-218)
    assert vcfg["class"] == "ValueNorm1" and algo["normalize_advantages"] is True
    self = types.SimpleNamespace(normalize_advantages=True, value_normalizer=getattr(valuenorm, vcfg["class"])(input_shape=(1,), **vcfg["kwargs"]))
    g = torch.Generator().manual_seed(seed)
    out = {}
    for r in range(rollouts):
        # GAE outputs in shape and scale: advantages a few units wide with an offset, returns around the episode's running reward (tens)
        adv = torch.randn(E, T, A, 1, generator=g) * (1.5 + r) + 0.3 * (r - 1)
        ret = torch.randn(E, T, A, 1, generator=g) * 4.0 - 12.0 + 5.0 * r
        tensordict = {"advantages": adv.clone(), "returns": ret.clone()}
        ns = {"self": self, "tensordict": tensordict, "torch": torch}
        for code in blocks:
            exec(code, ns)
        vn = self.value_normalizer
        out[f"r{r}_adv"], out[f"r{r}_ret"] = adv, ret
        out[f"r{r}_adv_normalised"], out[f"r{r}_ret_normalised"] = tensordict["advantages"], tensordict["returns"]
        out[f"r{r}_running_mean"], out[f"r{r}_running_mean_sq"] = vn.running_mean.clone(), vn.running_mean_sq.clone()
        out[f"r{r}_debiasing_term"] = vn.debiasing_term.clone()
        out[f"r{r}_denormalised_probe"] = vn.denormalize(torch.linspace(-2, 2, 9).unsqueeze(-1))      # mappo.py:378-379 on the next rollout
    out["meta"] = np.array([E, T, A, rollouts], dtype=np.int64)
    out["beta"] = np.float64(vcfg["kwargs"]["beta"])
    save("g_learner_moments", **out)


if __name__ == "__main__":
    gen_genbuffer()
    gen_learner_moments()
