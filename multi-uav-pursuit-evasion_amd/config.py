"""Configuration: the reference's `cfg/task/HideAndSeek*.yaml` schema -> `hns_cfg`.

hydra/omegaconf are not required: a config is a nested mapping with attribute access
(`cfg.task.num_agents`, `cfg.env.num_envs`, `cfg.sim.dt`, `cfg.algo.use_TP_net`) exactly as the
reference env reads it (reference: omni_drones/envs/isaac_env.py:54-69,
omni_drones/envs/hide_and_seek/hideandseek.py:236-325,435-455).  User YAML files written for
the reference (task files or a composed train config) load unchanged via `load_cfg`.

Derived constants are evaluated with torch fp32 tensors using the same expression forms the
reference uses, so they round identically (e.g. `dt / tau` is `tau.reciprocal() * dt` in torch).
"""
import copy
import math

import torch
import yaml

from . import abi

# --------------------------------------------------------------------------------------------
# defaults (values of the reference's cfg/task/HideAndSeek.yaml, cfg/base/{env,sim}_base.yaml,
# robots/assets/usd/crazyflie.yaml, robots/config.py; restated as data)
# --------------------------------------------------------------------------------------------
DEFAULT_TASK = {
    "name": "HideAndSeek",
    "env": {"num_envs": 2048, "env_spacing": 5, "max_episode_length": 800, "min_episode_length": 50},
    "sim": {"dt": 0.01, "substeps": 1, "gravity": [0, 0, -9.81], "device": "cuda:0"},
    "drone_model": "Crazyflie",
    "force_sensor": False,
    "time_encoding": True,
    "action_transform": "PIDrate",
    "scenario_flag": "wall",
    "num_agents": 3,
    "use_eval": 0,
    "use_partial_obs": 1,
    "use_random_cylinder": 1,
    "use_deployment": 0,
    "history_step": 10,
    "future_predcition_step": 5,
    "window_step": 1,
    "use_obstacles": 0,
    "arena_size": 0.9,
    "max_height": 1.2,
    "v_drone": 1.0,
    "v_prey": 1.3,
    "dist_reward_coef": 1.0,
    "catch_reward_coef": 20.0,
    "detect_reward_coef": 0.0,
    "collision_coef": 100.0,
    "speed_coef": 10.0,
    "init_smoothness_coef": 0.0,
    "smooth_lr": 0.0,
    "max_smoothness_coef": 5.0,
    "catch_radius": 0.3,
    "drone_detect_radius": 100.0,
    "target_detect_radius": 100.0,
    "collision_radius": 0.07,
    "cylinder": {"size": 0.1, "fixed_num": None, "min_num": 4, "max_num": 5, "obs_max_cylinder": 3},
    # --- switches of this build (no key of the reference's task files) ---
    # pid_reset: who clears the body-rate controller's integrator / last body rate.
    #   "reference": only `reset_pid = tensordict['done']` at the beginning of a step (transforms.py:449-454); `_reset_idx` leaves the controller
    #                alone, so with a collector that resets done envs the state carries over into the next episode — as in the reference;
    #   "on_reset":  hns_reset zeroes it for the envs it resets and the step never does (rounds 1-3 of this build).
    "pid_reset": "reference",
    # reset_extra_step: 1 = `_reset_idx` ends with one physics step of the whole scene, no rotor forces (hideandseek.py:722-723); 0 = none
    "reset_extra_step": 1,
    # action_input (NOT a default here: see `resolve_action_input`): "policy" | "motor" — what ("agents","action") of a stepped tensordict holds
}

DEFAULT_ALGO = {"name": "mappo", "use_TP_net": 0, "train_every": 64}

CRAZYFLIE = {
    "name": "crazyflie",
    "target_clip": 1.0,
    "max_thrust_ratio": 0.9,
    "fixed_yaw": 0,
    "inertia": {"xx": 1.4e-5, "yy": 1.4e-5, "zz": 2.17e-5},
    "mass": 0.0321,
    "drag_coef": 0.0,
    "rotor_configuration": {
        "arm_lengths": [0.043] * 4,
        "directions": [-1.0, 1.0, -1.0, 1.0],
        "force_constants": [2.350347298350041e-08] * 4,
        "max_rotation_velocities": [2315] * 4,
        "moment_constants": [7.24e-10] * 4,
        "num_rotors": 4,
        "rotor_angles": [0.78539816, 2.35619449, 3.92699082, 5.49778714],
        "time_constant": 0.025,
    },
}

# lee_position_controller.py:446-452
PID_GAINS = {"kp": [250.0, 250.0, 120.0], "ki": [500.0, 500.0, 16.7], "kd": [2.5, 2.5, 0.0],
             "ilimit": [33.3, 33.3, 166.7], "outlimit": 2.0 ** 15 - 1.0}

# robots/config.py:32-38 (+ hideandseek.py:539 overrides max_linear_velocity with v_drone)
RIGID_PROPS = {"linear_damping": 0.2, "angular_damping": 0.2, "max_angular_velocity": 1000.0}

# integrator options that have no reference counterpart (PhysX ground plane -> inelastic clamp)
DEFAULT_PHYSICS = {"ground_clamp": 1}


def _scenario(flag, size, height):
    """Fixed placements of use_random_cylinder=0 (hideandseek.py:480-531, 633-682)."""
    s, h = size, 0.5 * height
    base_d = [[0.6, 0.0, 0.5], [0.8, 0.0, 0.5], [0.8, -0.2, 0.5], [0.8, 0.2, 0.5]]
    if flag == "empty":
        return [], base_d, [-0.8, 0.0, 0.5]
    if flag == "passage":
        cyl = [[0.0, 3 * s, h], [-2 * s, 3 * s, h], [2 * s, 3 * s, h], [2 * s, -2 * s, h], [-2 * s, -2 * s, h], [0.0, -2 * s, h]]
        return cyl, [[0.6, 0.0, 0.5], [0.8, 0.2, 0.5], [0.8, -0.2, 0.5], [0.8, 0.2, 0.5]], [0.0, 0.6, 0.5]
    if flag == "wall":
        cyl = [[0.0, 1.5 * s, h], [0.0, -1.5 * s, h], [0.0, 4.5 * s, h], [0.0, -4.5 * s, h]]
        return cyl, [[0.6, 0.4, 0.5], [0.6, 0.0, 0.5], [0.6, -0.4, 0.5], [0.8, 0.2, 0.5]], [-0.8, 0.0, 0.5]
    if flag == "random":
        cyl = [[0.6, 0.4, 0.6], [-0.6, 0.4, 0.6], [-0.2, 0.4, 0.6], [0.0, 0.2, 0.6], [-0.2, -0.4, 0.6], [0.0, -0.2, 0.6]]
        return cyl, base_d, [-0.8, 0.0, 0.5]
    if flag == "narrow_gap":
        cyl = [[3 * s, -3 * s, h], [3 * s, 3 * s, h], [-3 * s, 3 * s, h], [-3 * s, -3 * s, h], [0.0, 3 * s, h]]
        return cyl, [[0.0, 0.7, 0.5], [0.2, 0.7, 0.5], [-0.2, 0.7, 0.5], [0.8, 0.2, 0.5]], [-0.5, 0.2, 0.5]
    raise ValueError(f"unknown scenario_flag {flag!r}")


# --------------------------------------------------------------------------------------------
# attribute-access mapping (the subset of OmegaConf's DictConfig the reference env relies on)
# --------------------------------------------------------------------------------------------
class Cfg(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = Cfg(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = Cfg(v) if isinstance(v, dict) and not isinstance(v, Cfg) else v

    def __deepcopy__(self, memo):
        return Cfg(copy.deepcopy(dict(self), memo))


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def resolve_action_input(task):
    """What the env will be fed under ("agents","action"), decided from the SAME key the reference's script decides on (scripts/train.py:148-171:
    `cfg.task.action_transform`), so a task file and the script that reads it cannot disagree:

    * `task.action_input` given ("policy" | "motor"): taken as is.
    * absent, `action_transform` none / null: "policy" — nothing runs in front (train.py:172-173), the raw policy output arrives and hns_step runs
      tanh -> CTBR -> body-rate PID itself (cfg/task/HideAndSeek_hip.yaml).
    * absent, any other `action_transform` (the reference's own task files say `PIDrate`, cfg/task/HideAndSeek.yaml:16): "motor" — train.py puts that
      controller transform in front of the env, which replaces the action by four rotor commands (utils/torchrl/transforms.py:455-456); hns_step
      then starts at `_pre_sim_step` (hideandseek.py:725-744), as the reference's env does.

    `make_cfg` (this build's programmatic constructor: tests, tools, bench.py) states "policy" explicitly; `load_cfg` (a YAML written for the
    reference, or hydra's composed config) leaves the key to this rule."""
    v = task.get("action_input", None)
    if v is not None:
        v = str(v).lower()
        if v not in ("policy", "motor"):
            raise ValueError("task.action_input must be 'policy' (raw policy output; the controller is fused into the step) or "
                             "'motor' (rotor commands of the caller's own controller transform)")
        return v
    tr = task.get("action_transform", None)
    return "policy" if tr is None or str(tr).lower() == "none" else "motor"


def make_cfg(task=None, algo=None, headless=True, _from_yaml=False, **task_overrides):
    """Compose a train-style config {task, algo, env, sim, headless} from partial mappings.  Called directly (not through `load_cfg`) the env
    takes the raw policy action whatever `action_transform` says (`action_input: policy`, see `resolve_action_input`)."""
    t = copy.deepcopy(DEFAULT_TASK)
    _merge(t, task or {})
    _merge(t, task_overrides)
    if not _from_yaml:
        t.setdefault("action_input", "policy")
    a = copy.deepcopy(DEFAULT_ALGO)
    _merge(a, algo or {})
    cfg = {"task": t, "algo": a, "env": t["env"], "sim": t["sim"], "headless": headless,
           "physics": copy.deepcopy(DEFAULT_PHYSICS)}
    return Cfg(cfg)


def load_cfg(path, **task_overrides):
    """Load a reference-schema YAML: either a task file (cfg/task/HideAndSeek*.yaml: top-level
    `name`, `env`, ...) or a composed config with `task:`/`algo:` sections.  hydra `defaults:`
    lists are ignored; missing keys fall back to the reference's defaults."""
    with open(path) as f:
        raw = yaml.safe_load(f) or {}
    raw.pop("defaults", None)
    if "task" in raw and isinstance(raw["task"], dict):
        task = raw["task"]
        task.pop("defaults", None)
        for k in ("env", "sim"):
            if isinstance(raw.get(k), dict):
                task.setdefault(k, {})
                _merge(task[k], raw[k])
        return make_cfg(task, raw.get("algo") if isinstance(raw.get("algo"), dict) else None,
                        raw.get("headless", True), _from_yaml=True, **task_overrides)
    return make_cfg(raw, None, True, _from_yaml=True, **task_overrides)


# Hover task (reference cfg/task/Hover.yaml, omni_drones/envs/single/hover.py:76-146)
DEFAULT_HOVER_TASK = {
    "name": "Hover",
    "env": {"num_envs": 100, "env_spacing": 5, "max_episode_length": 500, "min_episode_length": 50},
    "sim": {"dt": 0.01, "substeps": 1, "gravity": [0, 0, -9.81], "device": "cuda:0"},
    "drone_model": "crazyflie", "force_sensor": False,
    "reward_action_smoothness_weight": 0.0, "reward_distance_scale": 10.0, "reward_v_scale": 0.0,
    "reward_acc_scale": 0.0, "reward_jerk_scale": 0.0, "linear_vel_max": 3.0, "linear_acc_max": 10.0,
    "omega": False, "motor": False, "time_encoding": True, "action_transform": "PIDrate",
    "add_noise": False, "action_filter": False, "latency": False, "action_noise": False,
}


def make_hover_cfg(task=None, **task_overrides):
    t = copy.deepcopy(DEFAULT_HOVER_TASK)
    _merge(t, task or {})
    _merge(t, task_overrides)
    t.setdefault("action_input", "policy")          # programmatic constructor: the raw policy action (see resolve_action_input)
    return Cfg({"task": t, "algo": copy.deepcopy(DEFAULT_ALGO), "env": t["env"], "sim": t["sim"], "headless": True,
                "physics": copy.deepcopy(DEFAULT_PHYSICS)})


def resolve_hover_cfg(cfg, env_index_offset=0):
    """(hns_cfg with the drone/controller/integrator constants, hns_hover_cfg) for the Hover task."""
    t = cfg.task
    for k in ("omega", "motor", "add_noise", "latency", "action_noise"):
        if t.get(k, False):
            raise NotImplementedError(f"Hover option task.{k}=true is not built (plumbing configuration only)")
    if resolve_action_input(t) == "motor":
        # the reference's Hover.yaml says `action_transform: PIDrate` too: scripts/train.py would put the torch controller in front of an env whose step runs the
        # controller itself.  HideAndSeek takes the motor commands then (hns_cfg.action_input); the plumbing task refuses instead of flying a double controller.
        raise NotImplementedError("Hover: task.action_transform = %r means the caller's controller transform feeds rotor commands, and the Hover step (plumbing configuration) "
                                  "only takes the raw policy action — set `action_transform: none` (or `action_input: policy` if nothing runs in front of the env)"
                                  % (t.get("action_transform", None),))
    if "randomization" in t:
        raise NotImplementedError("Hover domain randomization is not built")
    base = make_cfg({"num_agents": 1, "env": dict(cfg.env), "sim": dict(cfg.sim)})
    base["physics"] = cfg.get("physics", DEFAULT_PHYSICS)
    c = resolve_hns_cfg(base, env_index_offset=env_index_offset)
    c.max_lin_vel = 1000.0 * (1.0 - 1e-6)       # Hover keeps the default max_linear_velocity (robots/config.py:36)
    h = abi.HnsHoverCfg()
    h.reward_distance_scale, h.reward_v_scale = float(t.reward_distance_scale), float(t.reward_v_scale)
    h.reward_acc_scale, h.reward_jerk_scale = float(t.reward_acc_scale), float(t.reward_jerk_scale)
    h.linear_vel_max, h.linear_acc_max = float(t.linear_vel_max), float(t.linear_acc_max)
    h.alpha = 0.8
    h.target_pos[:] = [0.0, 0.0, 1.0]
    h.target_heading[:] = [1.0, 0.0, 0.0]
    h.pos_lo[:], h.pos_hi[:] = [-1.0, -1.0, 0.05], [1.0, 1.0, 2.0]
    h.rpy_lo[:] = (torch.tensor([-0.2, -0.2, 0.0]) * torch.pi).tolist()
    h.rpy_hi[:] = (torch.tensor([0.2, 0.2, 0.5]) * torch.pi).tolist()
    return c, h


# --------------------------------------------------------------------------------------------
# cfg -> hns_cfg
# --------------------------------------------------------------------------------------------
def _f32(x):
    return float(torch.tensor(x, dtype=torch.float32))


def resolve_hns_cfg(cfg, num_envs=None, env_index_offset=0, drone_params=None, write_critic_state=True):
    t = cfg.task
    use_obst = int(cfg.algo.get("use_TP_net", 0)) and int(t.get("use_obstacles", 0))
    if int(cfg.algo.get("use_TP_net", 0)) and abi.tp_frame_dim(int(t.num_agents), int(t.cylinder.max_num), use_obst) > 80:
        raise NotImplementedError("TP_net frame wider than 80 values (7 + 3 num_agents + 3 cylinder.max_num with "
                                  "task.use_obstacles=1) is not built: the HIP predictor holds five 16-wide operand chunks")
    if t.get("drone_model", "Crazyflie").lower() != "crazyflie":
        raise NotImplementedError("only drone_model=Crazyflie is on the hot path")
    if not t.get("time_encoding", True):
        raise NotImplementedError("time_encoding=false is not supported")
    p = drone_params or CRAZYFLIE
    rc = p["rotor_configuration"]
    c = abi.HnsCfg()
    c.abi_version = abi.HNS_ABI_VERSION
    E = int(num_envs if num_envs is not None else cfg.env.num_envs)
    A = int(t.num_agents)
    Cn = int(t.cylinder.max_num)
    K = int(t.cylinder.obs_max_cylinder)
    if not (1 <= A <= abi.HNS_MAX_AGENTS):
        raise ValueError(f"num_agents must be in [1,{abi.HNS_MAX_AGENTS}]")
    if not (1 <= Cn <= abi.HNS_MAX_CYLINDERS):
        raise ValueError(f"cylinder.max_num must be in [1,{abi.HNS_MAX_CYLINDERS}]")
    if not (1 <= K <= Cn):
        raise ValueError("cylinder.obs_max_cylinder must be in [1, cylinder.max_num]")
    c.num_envs, c.num_agents, c.num_cylinders, c.obs_max_cylinder = E, A, Cn, K
    # two-evader extension (BASELINE config 5, not in the reference): task.num_targets: 2
    NT = int(t.get("num_targets", 1))
    if NT not in (1, 2):
        raise ValueError("task.num_targets must be 1 (the reference) or 2 (extension)")
    c.num_targets = NT
    pid_reset = str(t.get("pid_reset", "reference"))
    if pid_reset not in ("reference", "on_reset"):
        raise ValueError("task.pid_reset must be 'reference' or 'on_reset'")
    c.pid_reset_on_reset = 1 if pid_reset == "on_reset" else 0
    c.reset_extra_step = 1 if int(t.get("reset_extra_step", 1)) else 0
    c.action_input = abi.HNS_ACTION_MOTOR if resolve_action_input(t) == "motor" else abi.HNS_ACTION_POLICY
    c.stats_stride = E
    c.tp_use_obstacles = 1 if use_obst else 0
    c.max_episode_length = int(cfg.env.max_episode_length)
    c.use_deployment = int(t.use_deployment)
    c.fixed_yaw = int(p["fixed_yaw"])
    c.ground_clamp = int(cfg.get("physics", {}).get("ground_clamp", 1))
    c.write_critic_state = int(bool(write_critic_state))
    c.env_index_offset = int(env_index_offset)
    dt = float(cfg.sim.dt)
    c.dt = dt
    g = cfg.sim.get("gravity", [0, 0, -9.81])
    c.gravity = abs(float(g[2])) if isinstance(g, (list, tuple)) else abs(float(g))
    c.arena_size, c.max_height = float(t.arena_size), float(t.max_height)
    c.cylinder_size, c.cylinder_height = float(t.cylinder.size), float(t.max_height)
    c.catch_radius = float(t.catch_radius)
    c.drone_detect_radius, c.target_detect_radius = float(t.drone_detect_radius), float(t.target_detect_radius)
    c.collision_radius = float(t.collision_radius)
    c.v_drone = float(t.v_drone)
    c.v_prey = float(t.v_drone) * float(t.v_prey)                       # hideandseek.py:263
    c.dist_reward_coef, c.catch_reward_coef = float(t.dist_reward_coef), float(t.catch_reward_coef)
    c.detect_reward_coef, c.collision_coef = float(t.detect_reward_coef), float(t.collision_coef)
    c.speed_coef = float(t.speed_coef)
    init_s = float(t.get("init_smoothness_coef", t.get("smoothness_coef", 0.0)))   # envgen yaml names it smoothness_coef
    c.smoothness_coef = min(float(t.get("max_smoothness_coef", 5.0)), init_s)      # update_epoch = 0
    c.mask_value, c.invalid_z = -5.0, -20.0
    c.grid_size = 2 * float(t.cylinder.size)
    c.grid_num = int(float(t.arena_size) * 2 / (2 * float(t.cylinder.size)))       # hideandseek.py:578-579
    if c.grid_num > 16:
        raise ValueError("arena/cylinder grid larger than 16x16 is not supported")
    c.arena_sq = float(t.arena_size) ** 2
    c.coll_drone_dist = 2.0 * float(t.collision_radius)
    c.boundary = float(t.arena_size) - 0.1
    # drone
    c.mass = float(p["mass"])
    c.inertia[:] = [float(p["inertia"]["xx"]), float(p["inertia"]["yy"]), float(p["inertia"]["zz"])]
    w = torch.as_tensor(rc["max_rotation_velocities"]).float()
    kf = w.square() * torch.as_tensor(rc["force_constants"])                        # rotor_group.py:41
    km = w.square() * torch.as_tensor(rc["moment_constants"])
    ang = torch.as_tensor(rc["rotor_angles"])
    arm = torch.as_tensor(rc["arm_lengths"])
    c.kf[:] = kf.tolist()
    c.km[:] = km.tolist()
    c.rotor_dir[:] = [float(x) for x in rc["directions"]]
    c.rotor_px[:] = (torch.cos(ang) * arm).float().tolist()
    c.rotor_py[:] = (torch.sin(ang) * arm).float().tolist()
    tau = torch.clamp(torch.as_tensor(rc["time_constant"]).float(), 0, 1)
    tau = dt / tau                                                                  # rotor_group.py:58-61
    c.tau_up = c.tau_down = float(tau)
    c.max_thrust_ratio, c.target_clip = float(p["max_thrust_ratio"]), float(p["target_clip"])
    gravity = torch.tensor(c.mass, dtype=torch.float32) * 9.81                      # multirotor.py:246
    c.hover_throttle = float(torch.sqrt(gravity / kf.sum()))                        # multirotor.py:647-648
    c.pid_kp[:], c.pid_ki[:], c.pid_kd[:] = PID_GAINS["kp"], PID_GAINS["ki"], PID_GAINS["kd"]
    c.pid_ilimit[:] = PID_GAINS["ilimit"]
    c.pid_outlimit = PID_GAINS["outlimit"]
    c.lin_damp_factor = max(0.0, 1.0 - dt * RIGID_PROPS["linear_damping"])
    c.ang_damp_factor = max(0.0, 1.0 - dt * RIGID_PROPS["angular_damping"])
    c.max_ang_vel = RIGID_PROPS["max_angular_velocity"]
    c.max_lin_vel = float(t.v_drone) * (1.0 - 1e-6)
    one = torch.tensor(1.0, dtype=torch.float32)
    c.inv_mass = float(one / torch.tensor(c.mass, dtype=torch.float32))
    c.inv_inertia[:] = [float(one / torch.tensor(x, dtype=torch.float32)) for x in c.inertia]
    c.inv_num_agents = float(one / torch.tensor(float(A), dtype=torch.float32))
    c.inv_max_episode_length = float(one / torch.tensor(float(c.max_episode_length), dtype=torch.float32))
    c.inv_dt = float(one / torch.tensor(dt, dtype=torch.float32))
    # reset distributions, hideandseek.py:283-313
    r = float(t.arena_size) / math.sqrt(2.0)
    c.drone_xy_lo[:], c.drone_xy_hi[:] = [0.1, -r + 0.1], [r - 0.1, r - 0.1]
    c.target_xy_lo[:], c.target_xy_hi[:] = [-r + 0.1, -r + 0.1], [-0.1, r - 0.1]
    c.z_lo, c.z_hi = float(t.max_height) / 2 - 0.1, float(t.max_height) / 2 + 0.1
    use_eval = int(t.use_eval)
    if use_eval:
        c.rpy_lo[:], c.rpy_hi[:] = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]
    else:
        c.rpy_lo[:] = (torch.tensor([-0.2, -0.2, 0.0]) * torch.pi).tolist()
        c.rpy_hi[:] = (torch.tensor([0.2, 0.2, 0.2]) * torch.pi).tolist()
    fixed_num = t.cylinder.get("fixed_num", None)
    c.cyl_fixed_num = -1 if fixed_num is None else int(fixed_num)
    c.cyl_min_num = int(t.cylinder.min_num)
    if c.cyl_fixed_num > Cn or c.cyl_min_num > Cn:
        raise ValueError("cylinder.fixed_num/min_num exceed cylinder.max_num")
    eval_xy = [[0.6, 0.0], [0.8, 0.0], [0.8, -0.2], [0.8, 0.2]]                     # hideandseek.py:620-628
    if int(t.use_random_cylinder):
        c.init_mode = abi.HNS_INIT_EVAL if use_eval else abi.HNS_INIT_RANDOM
        if use_eval and A > len(eval_xy):
            raise ValueError("use_eval places at most 4 drones (reference hideandseek.py:620-625)")
        for a in range(min(A, len(eval_xy))):
            c.fixed_drone_pos[a][0], c.fixed_drone_pos[a][1] = eval_xy[a]
        c.fixed_target_pos[0], c.fixed_target_pos[1] = -0.8, 0.0
        # free cells inside the disc minus the cells of the pursuers and the evader(s) must cover the cylinder slots (:112-113)
        half = c.grid_num // 2
        free = sum(1 for i in range(c.grid_num) for j in range(c.grid_num)
                   if math.sqrt((i - half) ** 2 + (j - half) ** 2) < half)
        if free - (A + (2 if int(t.get("num_targets", 1)) == 2 else 1)) < Cn:
            raise ValueError("Not enough available grid cells for cylinder.max_num")
    else:
        c.init_mode = abi.HNS_INIT_SCENARIO
        cyl, dpos, tpos = _scenario(t.scenario_flag, float(t.cylinder.size), float(t.max_height))
        if A > len(dpos):
            raise ValueError("fixed scenarios place at most 4 drones")
        if len(cyl) > Cn:
            raise ValueError(f"scenario {t.scenario_flag!r} needs cylinder.max_num >= {len(cyl)}")
        if len({tuple(dpos[a]) for a in range(A)}) < A:
            import warnings
            # kept as the reference has it (hideandseek.py:673-679: `passage` lists its second drone twice) — with all four drones the pair's
            # separation is 0 and the downwash term 0 / 0: NaN states from the first step on, there as here
            warnings.warn(f"scenario {t.scenario_flag!r} places two of its {A} drones on the same point (as the reference does): the state turns NaN",
                          RuntimeWarning, stacklevel=2)
        for a in range(A):
            c.fixed_drone_pos[a][:] = dpos[a]
        c.fixed_target_pos[:] = tpos
        for k in range(Cn):                                                         # hideandseek.py:456-460
            c.fixed_cyl_pos[k][:] = cyl[k] if k < len(cyl) else [k * 2 * float(t.cylinder.size), 0.0, -20.0]
        c.fixed_cyl_active = len(cyl)
    return c
