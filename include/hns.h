/*
 * hns.h — C ABI of the MI355X-native HideAndSeek environment step.
 *
 * The reference (thu-uav/Multi-UAV-pursuit-evasion) has no FFI: its environment is the Python
 * class `HideAndSeek(IsaacEnv)` and the physics is PhysX behind `omni.isaac.core`.  This
 * header is the boundary a maintainer binds *below* that Python class; each entry point names
 * the reference code it replaces (paths relative to the reference repo):
 *
 *   hns_step   <->  TransformedEnv.step = PIDRateController._inv_call
 *                   (omni_drones/utils/torchrl/transforms.py:425-459)
 *                   + IsaacEnv._step (omni_drones/envs/isaac_env.py:231-240)
 *                   = HideAndSeek._pre_sim_step (envs/hide_and_seek/hideandseek.py:725-744)
 *                   + sim.step() [PhysX; replaced by the documented integrator, DESIGN.md §A5]
 *                   + _compute_state_and_obs (:746-917) + _compute_reward_and_done (:919-1065)
 *   hns_reset  <->  IsaacEnv._reset (isaac_env.py:210-225) = HideAndSeek._reset_idx
 *                   (hideandseek.py:609-723) + MultirotorBase._reset_idx
 *                   (robots/drone/multirotor.py:635-650) + the reset-time obs pass
 *   hns_create <->  IsaacEnv.__init__/HideAndSeek.__init__ parameter capture
 *                   (isaac_env.py:54-151, hideandseek.py:236-325, 435-455)
 *
 * Conventions: plain C types only; every `float*`/`uint8_t*` in hns_buffers is a DEVICE
 * pointer owned by the caller (e.g. torch tensors) that must stay valid while bound;
 * functions return 0 on success and a negative hns_status on error, never throw, never
 * allocate device memory after hns_create, never synchronise a stream or the device (three documented exceptions: hns_bind's FIRST
 * call does one blocking 1 KB upload; hns_step_kernel_ms waits for its last sample; a configuration setter waits only if eight
 * earlier changes are still queued).  hns_step / hns_reset / hns_tp_observe / the setters are legal inside a stream capture (a configuration change made inside a capture takes one of 16 pinned images made by hns_create and keeps it for the env's lifetime; HNS_ERR_CONFIG once they are used up).
 * The configuration setters (hns_set_v_prey, hns_set_smoothness_coef, hns_set_phase_profile) and a repeated hns_bind change a
 * device-resident parameter block with ONE stream-ordered copy enqueued on the stream of the latest hns_step / hns_reset / hns_tp_observe call
 * (the null stream before the first): launches already enqueued there keep the old values, later launches and graph replays on
 * that stream see the new ones.  The HIP device current at the call must be the env's.  Quaternions are
 * (w,x,y,z) (omni_drones/utils/torch.py:62,125).  All tensors are C-contiguous fp32 unless noted.
 */
#ifndef HNS_H_
#define HNS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HNS_ABI_VERSION 5
#define HNS_MAX_AGENTS 7    /* pursuers per env: a workgroup is 64 envs = A pursuer waves + one env wave (<= 512 threads) */
#define HNS_MAX_CYLINDERS 16
#define HNS_NUM_STATS 24    /* hideandseek.py:400-425 */
#define HNS_SELF_DIM 20     /* state_self without TP prediction, hideandseek.py:856-863 */

typedef enum hns_status {
    HNS_OK = 0,
    HNS_ERR_INVALID_ARG = -1,
    HNS_ERR_NOT_BOUND = -2,
    HNS_ERR_DEVICE = -3,   /* HIP runtime error, see hns_last_error() */
    HNS_ERR_NO_DEVICE = -4,
    HNS_ERR_CONFIG = -5    /* e.g. fewer free grid cells than cylinders, hideandseek.py:112-113 */
} hns_status;

/* row index of each statistic in hns_buffers.stats ([HNS_NUM_STATS][E]); order = spec order */
typedef enum hns_stat {
    HNS_ST_SUCCESS = 0, HNS_ST_COLLISION, HNS_ST_BLOCKED, HNS_ST_DISTANCE_REWARD,
    HNS_ST_DISTANCE_PREDICTED_REWARD, HNS_ST_SPEED_REWARD, HNS_ST_COLLISION_REWARD,
    HNS_ST_COLLISION_WALL, HNS_ST_COLLISION_CYLINDER, HNS_ST_COLLISION_DRONE,
    HNS_ST_DETECT_REWARD, HNS_ST_CATCH_REWARD, HNS_ST_SMOOTHNESS_REWARD, HNS_ST_SMOOTHNESS_MEAN,
    HNS_ST_SMOOTHNESS_MAX, HNS_ST_FIRST_CAPTURE_STEP, HNS_ST_SUM_DETECT_STEP, HNS_ST_RETURN,
    HNS_ST_ACTION_ERROR_ORDER1_MEAN, HNS_ST_ACTION_ERROR_ORDER1_MAX, HNS_ST_TARGET_PREDICTED_ERROR,
    HNS_ST_DISTANCE_THRESHOLD_L, HNS_ST_OUT_OF_ARENA, HNS_ST_SMOOTHNESS_COEF
} hns_stat;

/* What hns_step's `action` is — i.e. on which side of the boundary the reference's action transform runs (scripts/train.py:165-171).
 *   HNS_ACTION_POLICY: the raw policy output (pre-tanh).  hns_step runs PIDRateController._inv_call (transforms.py:425-459) and the
 *                      body-rate PID (lee_position_controller.py:476-550) itself; `prev_action`, `action_error`, `pid_*` are its state / outputs.
 *   HNS_ACTION_MOTOR:  the four rotor commands that transform left in ("agents","action") (transforms.py:455-456: the caller keeps
 *                      `action_transform: PIDrate` and runs the torch controller in front, as the reference's own task file has it).  hns_step
 *                      starts at HideAndSeek._pre_sim_step (hideandseek.py:725-744): `action` goes straight to the rotors (A3); `prev_action` and
 *                      `action_error` are INPUTS — what the transform set under ("info","prev_action") / ("stats","action_error_order1"),
 *                      copied into the bound buffers by the caller (:729-731) — and are not written; `pid_integ`, `reset_pid`, `ctbr`, `target_rate`
 *                      are not touched (the caller's controller owns that state); of `pid_last_rate` only the line-of-sight column (w) is. */
typedef enum hns_action_input { HNS_ACTION_POLICY = 0, HNS_ACTION_MOTOR = 1 } hns_action_input;

/* How reset places bodies (hideandseek.py:613-689). */
typedef enum hns_init_mode {
    HNS_INIT_RANDOM = 0,   /* use_random_cylinder=1, use_eval=0: uniform boxes + grid cylinders */
    HNS_INIT_EVAL = 1,     /* use_random_cylinder=1, use_eval=1: fixed xy, random z, zero rpy */
    HNS_INIT_SCENARIO = 2  /* use_random_cylinder=0: fixed drone/target/cylinder positions */
} hns_init_mode;

/*
 * All scalars the step/reset math needs, resolved on the host from cfg/task/HideAndSeek*.yaml
 * and robots/assets/usd/crazyflie.yaml.  Derived values are computed by the host in fp32
 * exactly as the reference computes them (noted per field).
 */
typedef struct hns_cfg {
    int32_t abi_version;       /* = HNS_ABI_VERSION */
    int32_t num_envs;          /* E  (this process' shard) */
    int32_t num_agents;        /* A  cfg.task.num_agents */
    int32_t num_cylinders;     /* C  cfg.task.cylinder.max_num (slots incl. inactive) */
    int32_t obs_max_cylinder;  /* k  cfg.task.cylinder.obs_max_cylinder (<= C) */
    int32_t max_episode_length;
    int32_t use_deployment;    /* smoothness reward on/off, hideandseek.py:993 */
    int32_t fixed_yaw;         /* crazyflie.yaml:6 */
    int32_t ground_clamp;      /* integrator: inelastic ground plane at z=0 */
    int32_t write_critic_state;/* also fill buffers.state_drones */
    int32_t init_mode;         /* hns_init_mode */
    int32_t cyl_min_num;       /* cylinder.min_num */
    int32_t cyl_fixed_num;     /* cylinder.fixed_num, -1 = null */
    int32_t grid_num;          /* int(arena*2/(2*size)) = 9, hideandseek.py:579 */
    int32_t env_index_offset;  /* global index of local env 0 (multi-GPU shard); keys the reset RNG */
    int32_t num_targets;       /* evaders per env: 0 or 1 = the reference (one evader); 2 = two-evader extension (below) */

    float dt;                  /* cfg.sim.dt */
    float gravity;             /* 9.81 */
    /* task */
    float arena_size, max_height, cylinder_size, cylinder_height;
    float catch_radius, drone_detect_radius, target_detect_radius, collision_radius;
    float v_drone;             /* speed-penalty threshold AND PhysX max_linear_velocity (hideandseek.py:539) */
    float v_prey;              /* v_drone * cfg.task.v_prey (hideandseek.py:263); runtime-updatable */
    float dist_reward_coef, catch_reward_coef, detect_reward_coef, collision_coef, speed_coef;
    float smoothness_coef;     /* min(max_smoothness_coef, init + smooth_lr*update_epoch), :988-989 */
    float mask_value;          /* -5, hideandseek.py:305 */
    float invalid_z;           /* -20, hideandseek.py:451 */
    float grid_size;           /* 2*cylinder_size */
    float arena_sq;            /* fp32(arena_size**2): python-double square, then cast (hideandseek.py:1096,979) */
    float coll_drone_dist;     /* fp32(2.0*collision_radius) (hideandseek.py:973) */
    float boundary;            /* arena_size - 0.1 */
    /* drone (Crazyflie) */
    float mass;
    float inertia[3];          /* diagonal */
    float kf[4];               /* max_rot_vel^2 * force_constant, rotor_group.py:41 */
    float km[4];               /* max_rot_vel^2 * moment_constant, rotor_group.py:42 */
    float rotor_dir[4];        /* directions */
    float rotor_px[4];         /* arm_length*cos(angle): torque_y = -sum(px*T) */
    float rotor_py[4];         /* arm_length*sin(angle): torque_x =  sum(py*T) */
    float tau_up, tau_down;    /* dt / clamp(time_constant,0,1), rotor_group.py:58-61 */
    float max_thrust_ratio, target_clip;
    float hover_throttle;      /* sqrt(m*g / sum(kf)), multirotor.py:647-648 */
    float pid_kp[3], pid_ki[3], pid_kd[3], pid_ilimit[3], pid_outlimit; /* lee_position_controller.py:446-452 */
    /* integrator (PhysX-like) */
    float lin_damp_factor;     /* max(0, 1 - dt*linear_damping)  (robots/config.py:32) */
    float ang_damp_factor;     /* max(0, 1 - dt*angular_damping) (robots/config.py:34) */
    float max_ang_vel;         /* 1000 rad/s (robots/config.py:38) */
    float inv_mass;            /* fp32(1/mass), fp32(1/inertia): the integrator multiplies by reciprocals */
    float inv_inertia[3];
    float inv_num_agents;      /* 1.0f/A: torch's CUDA mean multiplies the sum by 1/N (ReduceMomentKernel.cu) */
    float inv_max_episode_length; /* 1.0f/max_len: CUDA `tensor / python_scalar` multiplies by the fp32 reciprocal */
    float max_lin_vel;         /* v_drone*(1-1e-6): PhysX max_linear_velocity (hideandseek.py:539), set a hair
                                  inside so the clamped speed never trips `speed > v_drone` (:952) by rounding */
    float inv_dt;              /* fp32(1/fp32(dt)): `(rate - last) / self.dt` (lee_position_controller.py:509) on CUDA multiplies by the
                                  fp32 reciprocal of the Python scalar (BinaryDivTrueKernel.cu) */
    /* reset distributions (hideandseek.py:283-313) */
    float drone_xy_lo[2], drone_xy_hi[2], target_xy_lo[2], target_xy_hi[2];
    float z_lo, z_hi;
    float rpy_lo[3], rpy_hi[3];
    /* fixed placements for HNS_INIT_EVAL (xy only) / HNS_INIT_SCENARIO (hideandseek.py:480-531,633-682) */
    float fixed_drone_pos[HNS_MAX_AGENTS + 1][3];
    float fixed_target_pos[3];
    float fixed_cyl_pos[HNS_MAX_CYLINDERS][3];
    int32_t fixed_cyl_active;  /* HNS_INIT_SCENARIO: number of active cylinders */
    int32_t tp_use_obstacles;  /* task.use_obstacles: the predictor's frame also holds [x, y, cylinder_size] of every cylinder slot
                                  (hideandseek.py:808-816); 0 = the reference's default */
    int32_t pid_reset_on_reset;/* 0 = the reference: `_reset_idx` (hideandseek.py:609-723) never touches the body-rate controller; its integrator and
                                  last body rate are cleared only through `reset_pid` at the next step (buffers.reset_pid).  1 = hns_reset also zeroes
                                  pid_integ / pid_last_rate of the envs it resets (a fresh controller per episode; rounds 1-3 of this build) */
    int32_t stats_stride;      /* floats between consecutive rows of buffers.stats: 0 = num_envs (a [HNS_NUM_STATS, E] array of its own); a larger
                                  value lets an env over a SLICE of a bigger batch address its columns of that batch's array in place (the Python
                                  env steps the two halves of its batch on two streams when the predictor is on) */
    int32_t reset_extra_step;  /* 1 = the reference: `_reset_idx` ends with one `sim.step()` of the WHOLE scene (hideandseek.py:722-723) — every drone of
                                  every env (reset or not) integrates one dt with no rotor force (gravity + damping), every evader moves one dt with
                                  the velocity it holds; then the observation of all envs is recomputed (isaac_env.py:221).  0 = no extra step */
    int32_t action_input;      /* hns_action_input: what `action` of hns_step holds (below) */
} hns_cfg;

/*
 * Two-evader extension (num_targets = 2; NOT in the reference — BASELINE config 5 "6-pursuer/2-evader"):
 * each evader runs the reference's potential-field policy (hideandseek.py:1067-1141) on its own against all
 * pursuers, the arena and the cylinders (evaders ignore each other); line of sight, detection and masking
 * are evaluated per evader; the distance reward refers to the NEAREST evader, the catch reward to ANY
 * evader captured by any pursuer; `blocked` counts steps in which no pursuer sees either evader.  Shapes:
 * target_pos / target_vel [E,2,3]; obs_self / state_drones rows have 24 values = the reference's 20, the
 * relative position of evader 1, one zero; detect[e] is a bit mask (bit k = evader k detected); task vectors
 * are [drones | evader 0 | evader 1 | cylinders].  The predictor (hns_tp_*) runs the SAME network once per evader: unit u = 2 e + j
 * sees evader j's position / velocity under detection bit j; history / pred / groundtruth / tp_done are [2E, ...] (unit-major), the rows
 * have 24 + 6F values = [the reference's 20 + 3F row for evader 0 | relative position of evader 1, 0 | drone - predicted evader 1 (3F)].
 */
/* Device buffers, caller-owned.  Shapes in brackets; E,A,C,k as in hns_cfg. */
typedef struct hns_buffers {
    /* persistent state, updated in place */
    float *drone_state;    /* [E,A,13] pos3 quat4 linvel3 angvel3, world frame == info.drone_state */
    float *throttle;       /* [E,A,4]  rotor throttle, multirotor.py:216 */
    float *pid_integ;      /* [E,A,4]  xyz + pad, lee_position_controller.py:497-502 */
    float *pid_last_rate;  /* [E,A,4]  xyz + the pursuer's line-of-sight flag(s) in the state the buffers hold: 1 = the line to the evader
                            *          is blocked by a cylinder (+2 = to the second evader).  Derived state: written by step and reset
                            *          from the observation pass (hideandseek.py:786), read by the next step as the evader policy's
                            *          test (:1080) — same positions, same result; hns_set_state recomputes it from what it uploads. */
    float *prev_action;    /* [E,A,4]  == info.prev_action (ctbr of the last step) */
    float *target_pos;     /* [E,3]    evader position ([E,2,3] with num_targets = 2) */
    float *target_vel;     /* [E,3]    evader linear velocity set this step (hideandseek.py:741) */
    float *cylinders;      /* [E,C,3]  z<0 => inactive */
    float *progress;       /* [E]      float step counter, isaac_env.py:142-147 */
    float *stats;          /* [HNS_NUM_STATS,E] */
    /* per-step outputs */
    float *obs_self;       /* [E,A,20]       agents.observation.state_self */
    float *obs_others;     /* [E,A,A-1,3]    agents.observation.state_others (unused when A==1) */
    float *obs_cylinders;  /* [E,A,k,5]      agents.observation.cylinders == agents.state.cylinders */
    float *state_drones;   /* [E,A,20]       agents.state.state_drones; may be NULL if !write_critic_state */
    float *reward;         /* [E,A]          agents.reward */
    float *action_error;   /* [E,A]          stats.action_error_order1 (transforms.py:441) */
    uint8_t *done;         /* [E]            bool */
    uint8_t *detect;       /* [E]            bool, nullable: broadcast_detect (hideandseek.py:791), used by the TP_net input */
    /* failure detection (nullable): one sticky word, OR-ed by hns_step on the device, never cleared by the library.
     * bit 0: some pursuer's new rigid state is not finite; bit 1: some evader's new position; bit 2: some reward.
     * "Not finite" is decided on the left-to-right fp32 sum s of the values concerned: (s - s) != 0. */
    uint32_t *nonfinite;   /* [1] */
    /* optional outputs (nullable): the two extra keys PIDRateController._inv_call leaves on the tensordict (transforms.py:456-457) */
    float *ctbr;           /* [E,A,4]        controller output (roll, pitch, yaw command, thrust), lee_position_controller.py:548 */
    float *target_rate;    /* [E,A,4]        target body rate in deg/s (x, y, z, 0), transforms.py:447 */
    /* optional INPUT of hns_step (nullable): `reset_pid = tensordict['done']` (transforms.py:449-454 -> lee_position_controller.py:497-502) —
     * envs whose byte is non-zero start the step with pid_integ = pid_last_rate = 0.  It is read at the very beginning of the step and may ALIAS
     * `done` (written at its very end): the step then consumes the `done` its predecessor (or a reset, which clears it) left — what the root `done`
     * of a stepped tensordict holds in the reference's collector / rollout loops.  NULL = never reset through the step. */
    const uint8_t *reset_pid; /* [E] */
} hns_buffers;

/*
 * Hover task (BASELINE config 1, reference omni_drones/envs/single/hover.py): one Crazyflie per env,
 * 20-dim observation, position/heading/uprightness reward.  Plumbing-scale (tens of envs), so the
 * entry points are stateless: cfg (drone + sim fields of hns_cfg; num_agents = 1) and buffers per call.
 */
#define HNS_HOVER_NUM_STATS 39  /* hover.py:238-278, spec order */
#define HNS_HOVER_NUM_ACC 12    /* *_episode accumulators and last_* values, hover.py:150-155,313-320 */
typedef struct hns_hover_cfg {
    float reward_distance_scale, reward_v_scale, reward_acc_scale, reward_jerk_scale;
    float linear_vel_max, linear_acc_max;
    float alpha;                /* 0.8, hover.py:148 */
    float target_pos[3];        /* (0,0,1), hover.py:146 */
    float target_heading[3];    /* quat_axis(target_rot, 0) with target rpy = 0 -> (1,0,0), hover.py:301-303 */
    float pos_lo[3], pos_hi[3]; /* hover.py:129-132 */
    float rpy_lo[3], rpy_hi[3]; /* hover.py:137-140 */
} hns_hover_cfg;
typedef struct hns_hover_buffers {
    float *drone_state;   /* [E,1,13] */
    float *throttle, *pid_integ, *pid_last_rate, *prev_action;   /* [E,1,4] */
    float *progress;      /* [E] */
    float *stats;         /* [HNS_HOVER_NUM_STATS,E] */
    float *acc;           /* [HNS_HOVER_NUM_ACC,E] */
    float *obs;           /* [E,1,20] */
    float *reward;        /* [E,1] */
    uint8_t *done;        /* [E] */
} hns_hover_buffers;
/* One Hover step = PIDRateController._inv_call + Hover._pre_sim_step + integrator + _compute_state_and_obs
 * + _compute_reward_and_done (hover.py:322-523).  `cfg` supplies num_envs, max_episode_length, dt and the
 * drone/controller/integrator constants. */
int hns_hover_step(const hns_cfg *cfg, const hns_hover_cfg *hover, const hns_hover_buffers *buffers,
                   const float *action, void *stream);
/* Hover._reset_idx (hover.py:285-320) for the masked envs (NULL = all) + their observation. */
int hns_hover_reset(const hns_cfg *cfg, const hns_hover_cfg *hover, const hns_hover_buffers *buffers,
                    const uint8_t *reset_mask, uint64_t seed, uint32_t epoch, void *stream);

typedef struct hns_env hns_env;

/* Validate cfg, select the kernel specialisation, allocate nothing on the device.  The env belongs to the HIP
 * device that is current at this call: bind buffers of that device and launch with it current. */
int hns_create(const hns_cfg *cfg, hns_env **out);
void hns_destroy(hns_env *env);

/* Attach caller-owned device buffers (may be called again to re-point).  Host pointers and memory of another
 * GPU are refused here (HNS_ERR_INVALID_ARG) rather than faulting in a kernel; alignment is checked too. */
int hns_bind(hns_env *env, const hns_buffers *buffers);

/* One environment step for all E envs.  `action` = raw policy output [E,A,4] (pre-tanh) — or, with cfg.action_input =
 * HNS_ACTION_MOTOR, the rotor commands of the caller's own controller transform (hns_action_input above) —
 * device pointer.  `stream` is a hipStream_t (NULL = default stream).  Asynchronous. */
int hns_step(hns_env *env, const float *action, void *stream);

/* Reset the envs whose reset_mask byte is non-zero (NULL = all) and recompute their
 * observation.  `reset_mask` is a device pointer [E] (e.g. buffers.done).  Random draws come
 * from Philox4x32-10 keyed by (seed, global env index, reset epoch); the epoch is a host
 * counter advanced by every call.  Asynchronous. */
int hns_reset(hns_env *env, const uint8_t *reset_mask, uint64_t seed, void *stream);

/* envgen reset (hideandseek_envgen.py:875-902): like hns_reset, but the masked envs with index >=
 * task_first take their placement from `tasks` (device pointer, [E, 3A+3+3C] rows = drone positions,
 * evader position, cylinder positions — the reference's task vector; [E, 3A+6+3C] with both evaders'
 * positions in the two-evader extension) instead of sampling it;
 * orientations are still drawn from the Philox stream.  Masked envs < task_first reset as in hns_reset and
 * their rows of `tasks` are WRITTEN: the placement as sampled, before the extra physics step of
 * cfg.reset_extra_step (the reference archives `tasks_unif` as sampled, :883-895, and steps the scene
 * afterwards, :1013).  Rows of envs that are not masked are left alone. */
int hns_reset_tasks(hns_env *env, const uint8_t *reset_mask, float *tasks, int32_t task_first, uint64_t seed,
                    void *stream);

/* Extension, not in the reference: planar ray-fan range sensor on the bound state.
 * out: device pointer [E,A,num_rays]; ray r of a pursuer points along its horizontal heading rotated by
 * 2*pi*r/num_rays; range = distance to the first active cylinder or the arena wall, clamped to max_range. */
int hns_raycast(hns_env *env, int num_rays, float max_range, float *out, void *stream);

/*
 * Trajectory predictor in the observation (reference default `algo.use_TP_net: 1`).
 * Replaces the TP branch of HideAndSeek._compute_state_and_obs (hideandseek.py:805-854,871-880)
 * incl. the TP_net forward (learning/mappo.py:572-589: LSTM(I -> 64, 1 layer, zero initial state)
 * + Linear(64 -> 3F) + tanh) evaluated on a T-frame history, I = 7 + 3A (+ 3C with cfg.tp_use_obstacles):
 *   frame = [progress, evader pos (masked), evader vel (masked), pursuer positions]   (:815-820)
 *           + [x, y, cylinder_size] of every cylinder slot with task.use_obstacles     (:808-816)
 * I <= 80 (five 16-wide operand chunks): every shape the step kernels take (7 pursuers + 16 cylinder slots = 76 values).
 * The parameters are the caller's tensors in PyTorch layouts (the learner trains them,
 * scripts/train.py:180).  They are converted into a matrix-core operand image (`packed`) by
 * hns_tp_refresh: call it after every parameter update (hns_tp_bind schedules one).
 */
#define HNS_TP_HIDDEN 64           /* TP_net.hidden_dim, mappo.py:576 */
typedef struct hns_tp_buffers {
    const float *w_ih;        /* [4*64, I]   lstm.weight_ih_l0, gate order i,f,g,o */
    const float *w_hh;        /* [4*64, 64]  lstm.weight_hh_l0 */
    const float *b_ih;        /* [4*64]      lstm.bias_ih_l0 */
    const float *b_hh;        /* [4*64]      lstm.bias_hh_l0 */
    const float *w_fc;        /* [3F, 64]    fc.weight */
    const float *b_fc;        /* [3F]        fc.bias */
    void *packed;             /* [hns_tp_packed_bytes()] scratch, 16-byte aligned: operand image of the parameters */
    /* U = E units, or 2E with num_targets = 2 (unit 2 e + j = evader j of env e: the two-evader extension above) */
    float *history;           /* [U,T,I]  state: the sliding window == agents.TP.TP_input, oldest frame first */
    float *pred;              /* [U,F,3]  out: predicted evader positions, arena units (hideandseek.py:834-836) */
    float *obs_self;          /* [E,A,20+3F] out: agents.observation.state_self rows (:846-854); [E,A,24+6F] with two evaders */
    float *state_drones;      /* [E,A,20+3F] out, nullable: agents.state.state_drones (:873-880); [E,A,24+6F] with two evaders */
    float *groundtruth;       /* [U,3]    out: agents.TP.TP_groundtruth (:839-842) */
    uint8_t *tp_done;         /* [U]      out: agents.TP.TP_done (:838) */
} hns_tp_buffers;
size_t hns_tp_packed_bytes(void);
/* history_step T in [1,16], future_step F in [1,10]; max_episode_length <= 60000 (fp16-split operands). */
int hns_tp_bind(hns_env *env, const hns_tp_buffers *buffers, int32_t history_step, int32_t future_step);
/* Re-read the parameters (after an optimiser step / load_state_dict): one small launch on `stream`. */
int hns_tp_refresh(hns_env *env, void *stream);
/* Run after hns_step / hns_reset on the same stream: appends the frame of the bound step buffers to
 * the window (fill_history != 0: the window is filled with this frame, as the reference does on its
 * first call, hideandseek.py:825-828), evaluates TP_net, writes the 20+3F-value rows.  One launch; the kernel's workgroups serve 128, 64 or 32 units
 * depending on the batch size (small batches: more, shorter workgroups — the results do not depend on it; HNS_TP_TILES=1|2|4 forces one, A/B measurements). */
int hns_tp_observe(hns_env *env, int32_t fill_history, void *stream);

/*
 * Adaptive Environment Generator, device side (reference GenBuffer,
 * omni_drones/envs/hide_and_seek/hideandseek_envgen.py:209-377).
 */
/* Farthest-point sampling (replaces dgl.geometry.farthest_point_sampler, :291-304): out_idx[0] = start,
 * out_idx[r] = arg-max over all points of the minimum squared distance to out_idx[0..r-1] (ties -> lower
 * index).  points [n,d] fp32, out_idx [k] int32, scratch [hns_fps_scratch_bytes()] — device pointers.
 * One persistent launch (<= one workgroup per CU).  The XCD-local kernel (up to 36 coordinates, 131 072 points) accepts up to eight samples per
 * exchange — exactly those sequential sampling would select next, so out_idx does not depend on it.  If a workgroup never shows up the kernel gives up
 * instead of hanging: the first 8 bytes of scratch are then non-zero (check after synchronising). */
size_t hns_fps_scratch_bytes(void);
int hns_fps(const float *points, int32_t n, int32_t d, int32_t k, int32_t start, int32_t *out_idx, void *scratch, void *stream);
/* samplenearby (:316-370) with the grid sanity check (:187-207): tasks_out[t] = a random history entry,
 * pursuers / evader jittered by U(-1,1)*expand_step per coordinate (cylinders by {-1,0,1} cells when
 * expand_cylinders), clipped to the task bounds (:320-333); up to 10 attempts, then the entry itself.
 * history [n_hist, 3(A+1+C)], tasks_out [n_tasks, 3(A+1+C)] (3(A+2+C) with two evaders, both jittered like
 * pursuers): device pointers; Philox stream (seed, task). */
int hns_perturb_tasks(hns_env *env, const float *history, int32_t n_hist, float *tasks_out, int32_t n_tasks,
                      int32_t expand_cylinders, float expand_step, uint64_t seed, void *stream);

/* Curriculum hook (hideandseek.py:1012-1015): change the evader speed. */
int hns_set_v_prey(hns_env *env, float v_prey);
/* Smoothness schedule hook (hideandseek.py:988-991). */
int hns_set_smoothness_coef(hns_env *env, float coef);
/* Reset epoch (for checkpoint/resume and tests). */
int hns_set_reset_epoch(hns_env *env, uint32_t epoch);
uint32_t hns_get_reset_epoch(const hns_env *env);

/* Fixture injection / read-back for parity tests and checkpoints (the reference's counterpart is the
 * PhysX tensor view API, omni_drones/views/rigid_prim_view.py:61-118): `host` holds HOST pointers with the
 * shapes of hns_buffers; null fields are skipped; copies are asynchronous on `stream` (synchronise before
 * reading what hns_get_state wrote).  Equivalent to the caller copying into / out of its own bound buffers.
 * `host->stats` is always this handle's dense [HNS_NUM_STATS, num_envs]; for a handle over a slice of a larger batch
 * (cfg.stats_stride > num_envs) its columns of every device row are copied (one pitched copy). */
int hns_set_state(hns_env *env, const hns_buffers *host, void *stream);
int hns_get_state(hns_env *env, const hns_buffers *host, void *stream);
/* Recomputes the derived part of the state (the line-of-sight column of pid_last_rate, see hns_buffers) from drone_state, target_pos and
 * cylinders as the bound buffers hold them.  For callers that write positions into those buffers themselves instead of going through
 * hns_reset / hns_set_state (a checkpoint restored with tensor copies); hns_set_state runs it on what it uploads. */
int hns_refresh_derived_state(hns_env *env, void *stream);

/* Kernel timing: hns_enable_timing(env, n) times every n-th hns_step launch with a start / stop hipEvent pair bound
 * to that dispatch (hipExtLaunchKernelGGL: the timestamps of the kernel itself, on the launch stream; n = 0
 * disables; not for use inside a stream capture).  hns_step_kernel_ms returns the average device time (ms)
 * of the sampled launches since the last call (synchronises on the last sample); <0 if none. */
int hns_enable_timing(hns_env *env, int every_n);
float hns_step_kernel_ms(hns_env *env, int *num_launches);
/* Region timing (bench.py's roofline): ONE start event recorded on `stream` by hns_region_begin, one stop event by hns_region_end — no
 * per-launch host cost in between; hns_region_ms waits for the stop event and returns the device time between the two (ms; <0 without a
 * complete pair).  Divided by the launches in between it is the step kernel's duration INCLUDING the gap to its successor. */
int hns_region_begin(hns_env *env, void *stream);
int hns_region_end(hns_env *env, void *stream);
float hns_region_ms(hns_env *env);
/* Measurement yardstick (SURVEY §8d "achievable with a device copy kernel"): dst[i] = src[i] over `bytes` (a multiple of 16, both 16-byte
 * aligned device pointers) as 16-byte loads / stores, one float4 per thread.  Not part of the environment. */
int hns_copy_f4(void *dst, const void *src, size_t bytes, void *stream);

/* Measurement: the shader clock the chip is running at.  One wave spins for `ticks` periods of the constant 100 MHz clock and writes out[0] = shader-clock
 * cycles elapsed, out[1] = 100 MHz ticks elapsed (device pointer, 2 x uint64): MHz = 100 * out[0] / out[1].  MI355X clocks to its power budget; bench.py
 * reports this before and after its timed region.  Not part of the environment. */
int hns_clock_probe(unsigned long long *out, uint32_t ticks, void *stream);

/* Diagnostics: attach a device buffer of [num_waves, 16] uint64 (num_waves = ceil(E/64)*(A+1); ceil(E/64)*(2A+1) when hns_step_mapping() is 1);
 * lane 0 of every wave of the step kernel then stamps the shader clock at up to 16 phase boundaries (NULL detaches). */
int hns_set_phase_profile(hns_env *env, unsigned long long *device_buf);

/* Which mapping of the step kernel serves this env: 0 = 64-env tiles of A + 1 waves (batches that fill the chip), 1 = the small-batch mapping
 * (2 A + 1 waves per tile: a helper wave per pursuer wave; one evader, E % 64 == 0, obs_max_cylinder <= 4, at most two tiles per compute
 * unit — one with four and more pursuers).  Same buffers, bit for bit, either way; chosen by hns_create (the environment variable HNS_STEP_MAPPING=tile|small overrides it
 * where the shape allows both — A/B measurements and tests; HNS_STEP_PRIO=0|1 likewise overrides whether the tile mapping's pursuer waves
 * start at a raised issue priority, which hns_create decides from the shape and has no effect on any result). */
int hns_step_mapping(const hns_env *env);

/* The per-rollout moments of the data-parallel advantage normalisation (learning/mappo.py:391-396 made data-parallel; sharding.py) in ONE launch:
 * out[0..4] = [sum v, sum v^2, n, sum s, m] in fp64 over `values` [n] fp32 and `success` [m] fp32 (m may be 0: success NULL) — device
 * pointers; one workgroup, fixed summation order (the same inputs give the same bits on every run). */
int hns_moments(const float *values, int64_t n, const float *success, int64_t m, double *out, void *stream);
/* The same launch with ValueNorm1's batch moments riding along (learning/utils/valuenorm.py:83-91: `input_vector.mean()`, `(input_vector**2).mean()` over the
 * rollout's returns, mappo.py:398-399) — SURVEY §8(e)(3): out[0..7] = [sum adv, sum adv^2, n, sum success, m, sum ret, sum ret^2, n_returns]. */
int hns_rollout_moments(const float *advantages, int64_t n, const float *success, int64_t m, const float *returns, int64_t n_returns, double *out, void *stream);

int hns_abi_version(void);
size_t hns_cfg_size(void);   /* sizeof(hns_cfg) the library was built with (binding self-check) */
const char *hns_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* HNS_H_ */
