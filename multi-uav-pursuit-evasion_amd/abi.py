"""ctypes mirror of include/hns.h and loader of the HIP shared library (libhns.so).

The product path has no CPU fallback: `load_library()` raises if the HIP extension has not
been built (`python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

HNS_ABI_VERSION = 5
HNS_MAX_AGENTS = 7
HNS_MAX_CYLINDERS = 16
HNS_NUM_STATS = 24
HNS_SELF_DIM = 20

HNS_INIT_RANDOM, HNS_INIT_EVAL, HNS_INIT_SCENARIO = 0, 1, 2
HNS_ACTION_POLICY, HNS_ACTION_MOTOR = 0, 1      # hns_action_input

# row order of hns_buffers.stats == stats_spec order (reference hideandseek.py:400-425)
STAT_NAMES = [
    "success", "collision", "blocked", "distance_reward", "distance_predicted_reward",
    "speed_reward", "collision_reward", "collision_wall", "collision_cylinder", "collision_drone",
    "detect_reward", "catch_reward", "smoothness_reward", "smoothness_mean", "smoothness_max",
    "first_capture_step", "sum_detect_step", "return", "action_error_order1_mean",
    "action_error_order1_max", "target_predicted_error", "distance_threshold_L", "out_of_arena",
    "smoothness_coef",
]
assert len(STAT_NAMES) == HNS_NUM_STATS

_f = C.c_float
_i = C.c_int32


class HnsCfg(C.Structure):
    _fields_ = [
        ("abi_version", _i), ("num_envs", _i), ("num_agents", _i), ("num_cylinders", _i),
        ("obs_max_cylinder", _i), ("max_episode_length", _i), ("use_deployment", _i),
        ("fixed_yaw", _i), ("ground_clamp", _i), ("write_critic_state", _i), ("init_mode", _i),
        ("cyl_min_num", _i), ("cyl_fixed_num", _i), ("grid_num", _i), ("env_index_offset", _i),
        ("num_targets", _i),
        ("dt", _f), ("gravity", _f),
        ("arena_size", _f), ("max_height", _f), ("cylinder_size", _f), ("cylinder_height", _f),
        ("catch_radius", _f), ("drone_detect_radius", _f), ("target_detect_radius", _f),
        ("collision_radius", _f), ("v_drone", _f), ("v_prey", _f),
        ("dist_reward_coef", _f), ("catch_reward_coef", _f), ("detect_reward_coef", _f),
        ("collision_coef", _f), ("speed_coef", _f), ("smoothness_coef", _f),
        ("mask_value", _f), ("invalid_z", _f), ("grid_size", _f), ("arena_sq", _f),
        ("coll_drone_dist", _f), ("boundary", _f),
        ("mass", _f), ("inertia", _f * 3), ("kf", _f * 4), ("km", _f * 4), ("rotor_dir", _f * 4),
        ("rotor_px", _f * 4), ("rotor_py", _f * 4), ("tau_up", _f), ("tau_down", _f),
        ("max_thrust_ratio", _f), ("target_clip", _f), ("hover_throttle", _f),
        ("pid_kp", _f * 3), ("pid_ki", _f * 3), ("pid_kd", _f * 3), ("pid_ilimit", _f * 3),
        ("pid_outlimit", _f),
        ("lin_damp_factor", _f), ("ang_damp_factor", _f), ("max_ang_vel", _f), ("inv_mass", _f), ("inv_inertia", _f * 3), ("inv_num_agents", _f),
        ("inv_max_episode_length", _f), ("max_lin_vel", _f), ("inv_dt", _f),
        ("drone_xy_lo", _f * 2), ("drone_xy_hi", _f * 2), ("target_xy_lo", _f * 2),
        ("target_xy_hi", _f * 2), ("z_lo", _f), ("z_hi", _f), ("rpy_lo", _f * 3), ("rpy_hi", _f * 3),
        ("fixed_drone_pos", (_f * 3) * (HNS_MAX_AGENTS + 1)), ("fixed_target_pos", _f * 3),
        ("fixed_cyl_pos", (_f * 3) * HNS_MAX_CYLINDERS), ("fixed_cyl_active", _i), ("tp_use_obstacles", _i),
        ("pid_reset_on_reset", _i), ("stats_stride", _i), ("reset_extra_step", _i), ("action_input", _i),
    ]

    def copy(self):
        out = HnsCfg()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(HnsCfg))
        return out


_fp = C.c_void_p  # device (or, for the oracle, host) pointers travel as integers

BUFFER_FIELDS = [
    "drone_state", "throttle", "pid_integ", "pid_last_rate", "prev_action", "target_pos",
    "target_vel", "cylinders", "progress", "stats", "obs_self", "obs_others", "obs_cylinders",
    "state_drones", "reward", "action_error", "done", "detect", "nonfinite", "ctbr", "target_rate", "reset_pid",
]
# nullable fields: ctbr / target_rate are bound only with task.publish_ctbr (transforms.py:456-457); reset_pid is an INPUT that may alias
# `done` (include/hns.h) and is never allocated by buffer_shapes
OPTIONAL_BUFFER_FIELDS = ("ctbr", "target_rate", "reset_pid")


class HnsBuffers(C.Structure):
    _fields_ = [(name, _fp) for name in BUFFER_FIELDS]


def self_dim(num_targets=1):
    """Values per state_self / state_drones row: the reference's 20; 24 with the two-evader extension."""
    return 24 if num_targets == 2 else HNS_SELF_DIM


def buffer_shapes(E, A, Cn, K, num_targets=1, publish_ctbr=False):
    """Shape (and dtype name) of every hns_buffers field (the optional ones only when asked for)."""
    tgt = (E, 2, 3) if num_targets == 2 else (E, 3)
    D = self_dim(num_targets)
    extra = {"ctbr": ((E, A, 4), "float32"), "target_rate": ((E, A, 4), "float32")} if publish_ctbr else {}
    return {**_required_buffer_shapes(E, A, Cn, K, tgt, D), **extra}


def _required_buffer_shapes(E, A, Cn, K, tgt, D):
    return {
        "drone_state": ((E, A, 13), "float32"), "throttle": ((E, A, 4), "float32"),
        "pid_integ": ((E, A, 4), "float32"), "pid_last_rate": ((E, A, 4), "float32"),
        "prev_action": ((E, A, 4), "float32"), "target_pos": (tgt, "float32"),
        "target_vel": (tgt, "float32"), "cylinders": ((E, Cn, 3), "float32"),
        "progress": ((E,), "float32"), "stats": ((HNS_NUM_STATS, E), "float32"),
        "obs_self": ((E, A, D), "float32"), "obs_others": ((E, A, max(A - 1, 0), 3), "float32"),
        "obs_cylinders": ((E, A, K, 5), "float32"), "state_drones": ((E, A, D), "float32"),
        "reward": ((E, A), "float32"), "action_error": ((E, A), "float32"), "done": ((E,), "uint8"),
        "detect": ((E,), "uint8"), "nonfinite": ((1,), "int32"),
    }


HNS_HOVER_NUM_STATS = 39
HNS_HOVER_NUM_ACC = 12
HOVER_STAT_NAMES = [
    "return", "pos_bonus", "head_bonus", "reward_pos", "reward_up", "reward_vel", "reward_acc", "reward_jerk",
    "episode_len", "pos_error", "heading_alignment", "uprightness", "action_smoothness", "linear_v_max",
    "angular_v_max", "linear_a_max", "angular_a_max", "linear_jerk_max", "angular_jerk_max", "linear_v_mean",
    "angular_v_mean", "linear_a_mean", "angular_a_mean", "linear_jerk_mean", "angular_jerk_mean", "motor1",
    "motor2", "motor3", "motor4", "cmd_r", "cmd_p", "cmd_y", "cmd_thrust", "target_r_rate", "target_p_rate",
    "target_y_rate", "real_r_rate", "real_p_rate", "real_y_rate",
]
assert len(HOVER_STAT_NAMES) == HNS_HOVER_NUM_STATS


class HnsHoverCfg(C.Structure):
    _fields_ = [("reward_distance_scale", _f), ("reward_v_scale", _f), ("reward_acc_scale", _f), ("reward_jerk_scale", _f),
                ("linear_vel_max", _f), ("linear_acc_max", _f), ("alpha", _f), ("target_pos", _f * 3),
                ("target_heading", _f * 3), ("pos_lo", _f * 3), ("pos_hi", _f * 3), ("rpy_lo", _f * 3), ("rpy_hi", _f * 3)]


HOVER_BUFFER_FIELDS = ["drone_state", "throttle", "pid_integ", "pid_last_rate", "prev_action", "progress", "stats",
                       "acc", "obs", "reward", "done"]


class HnsHoverBuffers(C.Structure):
    _fields_ = [(name, _fp) for name in HOVER_BUFFER_FIELDS]


def hover_buffer_shapes(E):
    return {"drone_state": ((E, 1, 13), "float32"), "throttle": ((E, 1, 4), "float32"), "pid_integ": ((E, 1, 4), "float32"),
            "pid_last_rate": ((E, 1, 4), "float32"), "prev_action": ((E, 1, 4), "float32"), "progress": ((E,), "float32"),
            "stats": ((HNS_HOVER_NUM_STATS, E), "float32"), "acc": ((HNS_HOVER_NUM_ACC, E), "float32"),
            "obs": ((E, 1, HNS_SELF_DIM), "float32"), "reward": ((E, 1), "float32"), "done": ((E,), "uint8")}


HNS_OK, HNS_ERR_INVALID_ARG, HNS_ERR_NOT_BOUND, HNS_ERR_DEVICE, HNS_ERR_NO_DEVICE, HNS_ERR_CONFIG = 0, -1, -2, -3, -4, -5

# ---- trajectory predictor (include/hns.h: hns_tp_buffers) ------------------------------------------------
HNS_TP_HIDDEN = 64
TP_PACKED_BYTES = 16 * max(2 * 8 * 4 * 64 + 2 * 8 * 3 * 64 + 2 * 4 * 64 + 8 * 2 * 16 // 4 + 2 * 16 // 4,   # hns_tp_packed_bytes(): the tile kernel's image at three frame chunks
                           8 * 2 * (5 + 4) * 64 + 2 * 4 * 64 + 256 // 4 + 32 // 4)                          # and the weight-stationary kernel's at five
TP_WEIGHT_FIELDS = ["w_ih", "w_hh", "b_ih", "b_hh", "w_fc", "b_fc"]
TP_BUFFER_FIELDS = TP_WEIGHT_FIELDS + ["packed", "history", "pred", "obs_self", "state_drones", "groundtruth", "tp_done"]
# hns_tp_buffers weight field -> TP_net.state_dict() key (learning/mappo.py:572-589)
TP_STATE_DICT_KEYS = {"w_ih": "lstm.weight_ih_l0", "w_hh": "lstm.weight_hh_l0", "b_ih": "lstm.bias_ih_l0",
                      "b_hh": "lstm.bias_hh_l0", "w_fc": "fc.weight", "b_fc": "fc.bias"}


class HnsTpBuffers(C.Structure):
    _fields_ = [(name, _fp) for name in TP_BUFFER_FIELDS]


def tp_frame_dim(A, C=0, use_obstacles=False):
    """Width of one predictor frame (hideandseek.py:808-820)."""
    return 7 + 3 * A + (3 * C if use_obstacles else 0)


def tp_buffer_shapes(E, A, T, F, I=None, num_targets=1):
    """Two evaders (extension): the predictor runs once per (env, evader) unit — window, prediction, ground truth and done flag are
    [E * 2, ...] (unit 2 e + j), rows carry both predictions (24 + 6F values)."""
    NT = 2 if num_targets == 2 else 1
    I, D = (7 + 3 * A if I is None else I), self_dim(NT) + 3 * F * NT
    U = E * NT
    return {"w_ih": ((4 * HNS_TP_HIDDEN, I), "float32"), "w_hh": ((4 * HNS_TP_HIDDEN, HNS_TP_HIDDEN), "float32"),
            "b_ih": ((4 * HNS_TP_HIDDEN,), "float32"), "b_hh": ((4 * HNS_TP_HIDDEN,), "float32"),
            "w_fc": ((3 * F, HNS_TP_HIDDEN), "float32"), "b_fc": ((3 * F,), "float32"),
            "packed": ((TP_PACKED_BYTES,), "uint8"), "history": ((U, T, I), "float32"), "pred": ((U, F, 3), "float32"), "obs_self": ((E, A, D), "float32"),
            "state_drones": ((E, A, D), "float32"), "groundtruth": ((U, 3), "float32"), "tp_done": ((U,), "uint8")}


_LIB = None
LIB_NAME = "libhns.so"


def library_path():
    """In-tree libhns.so; HNS_LIBRARY names another build of the same ABI (A/B measurement builds, tools/build_variant.sh)."""
    return os.environ.get("HNS_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


def load_library():
    """dlopen the HIP extension and declare its C-ABI. Raises if it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback for the product path.")
    lib = C.CDLL(path)
    lib.hns_create.argtypes = [C.POINTER(HnsCfg), C.POINTER(C.c_void_p)]
    lib.hns_create.restype = C.c_int
    lib.hns_destroy.argtypes = [C.c_void_p]
    lib.hns_destroy.restype = None
    lib.hns_bind.argtypes = [C.c_void_p, C.POINTER(HnsBuffers)]
    lib.hns_bind.restype = C.c_int
    lib.hns_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hns_step.restype = C.c_int
    lib.hns_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.hns_reset.restype = C.c_int
    lib.hns_reset_tasks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p]
    lib.hns_reset_tasks.restype = C.c_int
    lib.hns_raycast.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    lib.hns_raycast.restype = C.c_int
    lib.hns_set_v_prey.argtypes = [C.c_void_p, C.c_float]
    lib.hns_set_v_prey.restype = C.c_int
    lib.hns_set_smoothness_coef.argtypes = [C.c_void_p, C.c_float]
    lib.hns_set_smoothness_coef.restype = C.c_int
    lib.hns_set_reset_epoch.argtypes = [C.c_void_p, C.c_uint32]
    lib.hns_set_reset_epoch.restype = C.c_int
    lib.hns_get_reset_epoch.argtypes = [C.c_void_p]
    lib.hns_get_reset_epoch.restype = C.c_uint32
    lib.hns_enable_timing.argtypes = [C.c_void_p, C.c_int]
    lib.hns_enable_timing.restype = C.c_int
    lib.hns_step_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.hns_step_kernel_ms.restype = C.c_float
    lib.hns_region_begin.argtypes = [C.c_void_p, C.c_void_p]
    lib.hns_region_begin.restype = C.c_int
    lib.hns_region_end.argtypes = [C.c_void_p, C.c_void_p]
    lib.hns_region_end.restype = C.c_int
    lib.hns_region_ms.argtypes = [C.c_void_p]
    lib.hns_region_ms.restype = C.c_float
    lib.hns_moments.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.hns_moments.restype = C.c_int
    lib.hns_rollout_moments.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.hns_rollout_moments.restype = C.c_int
    lib.hns_clock_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.hns_clock_probe.restype = C.c_int
    lib.hns_copy_f4.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.hns_copy_f4.restype = C.c_int
    lib.hns_set_phase_profile.argtypes = [C.c_void_p, C.c_void_p]
    lib.hns_set_phase_profile.restype = C.c_int
    lib.hns_step_mapping.argtypes = [C.c_void_p]
    lib.hns_step_mapping.restype = C.c_int
    lib.hns_hover_step.argtypes = [C.POINTER(HnsCfg), C.POINTER(HnsHoverCfg), C.POINTER(HnsHoverBuffers), C.c_void_p, C.c_void_p]
    lib.hns_hover_step.restype = C.c_int
    lib.hns_hover_reset.argtypes = [C.POINTER(HnsCfg), C.POINTER(HnsHoverCfg), C.POINTER(HnsHoverBuffers), C.c_void_p,
                                    C.c_uint64, C.c_uint32, C.c_void_p]
    lib.hns_hover_reset.restype = C.c_int
    lib.hns_tp_bind.argtypes = [C.c_void_p, C.POINTER(HnsTpBuffers), C.c_int32, C.c_int32]
    lib.hns_tp_bind.restype = C.c_int
    lib.hns_tp_refresh.argtypes = [C.c_void_p, C.c_void_p]
    lib.hns_tp_refresh.restype = C.c_int
    lib.hns_tp_packed_bytes.argtypes = []
    lib.hns_tp_packed_bytes.restype = C.c_size_t
    lib.hns_tp_observe.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.hns_tp_observe.restype = C.c_int
    lib.hns_set_state.argtypes = [C.c_void_p, C.POINTER(HnsBuffers), C.c_void_p]
    lib.hns_set_state.restype = C.c_int
    lib.hns_get_state.argtypes = [C.c_void_p, C.POINTER(HnsBuffers), C.c_void_p]
    lib.hns_get_state.restype = C.c_int
    lib.hns_refresh_derived_state.argtypes = [C.c_void_p, C.c_void_p]
    lib.hns_refresh_derived_state.restype = C.c_int
    lib.hns_fps_scratch_bytes.argtypes = []
    lib.hns_fps_scratch_bytes.restype = C.c_size_t
    lib.hns_fps.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hns_fps.restype = C.c_int
    lib.hns_perturb_tasks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_void_p]
    lib.hns_perturb_tasks.restype = C.c_int
    lib.hns_abi_version.argtypes = []
    lib.hns_abi_version.restype = C.c_int
    lib.hns_cfg_size.argtypes = []
    lib.hns_cfg_size.restype = C.c_size_t
    lib.hns_last_error.argtypes = []
    lib.hns_last_error.restype = C.c_char_p
    if lib.hns_abi_version() != HNS_ABI_VERSION:
        raise RuntimeError("libhns.so ABI version mismatch")
    if lib.hns_tp_packed_bytes() != TP_PACKED_BYTES:
        raise RuntimeError("hns_tp_packed_bytes() differs from abi.TP_PACKED_BYTES")
    if lib.hns_cfg_size() != C.sizeof(HnsCfg):
        raise RuntimeError("hns_cfg layout mismatch between include/hns.h and abi.py")
    _LIB = lib
    return lib


EXPORTED_SYMBOLS = [
    "hns_create", "hns_destroy", "hns_bind", "hns_step", "hns_reset", "hns_reset_tasks", "hns_raycast", "hns_set_v_prey",
    "hns_set_smoothness_coef", "hns_set_reset_epoch", "hns_get_reset_epoch", "hns_enable_timing",
    "hns_step_kernel_ms", "hns_region_begin", "hns_region_end", "hns_region_ms", "hns_copy_f4", "hns_moments", "hns_rollout_moments", "hns_clock_probe", "hns_set_phase_profile", "hns_step_mapping", "hns_set_state", "hns_get_state", "hns_refresh_derived_state", "hns_fps", "hns_fps_scratch_bytes", "hns_perturb_tasks", "hns_tp_bind", "hns_tp_refresh", "hns_tp_packed_bytes", "hns_tp_observe", "hns_hover_step", "hns_hover_reset", "hns_abi_version", "hns_cfg_size", "hns_last_error",
]
