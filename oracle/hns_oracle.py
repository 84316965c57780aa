"""ctypes wrapper of oracle/libhns_oracle.so — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It reuses the ABI struct mirrors (hns_amd.abi) because the oracle operates on the same
hns_cfg/hns_buffers as the HIP library, with HOST pointers (numpy arrays).
"""
import ctypes as C
import os
import subprocess

import numpy as np

import hns_amd  # noqa: F401  (registers the package alias)
from hns_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    # HNS_ORACLE_SANITIZE=1: the AddressSanitizer + UBSan build (the process must run with libasan preloaded)
    name = "libhns_oracle_asan.so" if os.environ.get("HNS_ORACLE_SANITIZE") == "1" else "libhns_oracle.so"
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "hns_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "hns.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", name])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.hns_oracle_cfg_size.restype = C.c_size_t
        assert _LIB.hns_oracle_cfg_size() == C.sizeof(abi.HnsCfg), "hns_cfg layout mismatch"
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def alloc_buffers(cfg):
    """Zero-initialised host arrays for every hns_buffers field."""
    shapes = abi.buffer_shapes(cfg.num_envs, cfg.num_agents, cfg.num_cylinders, cfg.obs_max_cylinder, cfg.num_targets)
    return {k: np.zeros(shape, dtype=dt) for k, (shape, dt) in shapes.items()}


def as_struct(arrs):
    b = abi.HnsBuffers()
    for k in abi.BUFFER_FIELDS:
        a = arrs.get(k)
        if a is None:
            assert k in abi.OPTIONAL_BUFFER_FIELDS, k
            continue
        assert a.flags["C_CONTIGUOUS"]
        setattr(b, k, a.ctypes.data)
    return b


def set_threads(n):
    """Host threads for step() (envs are independent; results identical for any count)."""
    lib().hns_oracle_set_threads(int(n))


def step(cfg, arrs, action):
    """One step.  `reset_pid` (include/hns.h): an explicit `arrs["reset_pid"]` ([E] u8, or None for NULL) is passed on; without the key the
    wrapper mirrors the Python env — with cfg.pid_reset_on_reset == 0 (task.pid_reset: reference) the input aliases `done`."""
    action = f32(action)
    b = as_struct(arrs)
    if "reset_pid" not in arrs and not cfg.pid_reset_on_reset:
        b.reset_pid = arrs["done"].ctypes.data
    rc = lib().hns_oracle_step(C.byref(cfg), C.byref(b), _p(action))
    assert rc == 0, rc


def reset(cfg, arrs, mask, seed, epoch):
    b = as_struct(arrs)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    rc = lib().hns_oracle_reset(C.byref(cfg), C.byref(b), _p(m), C.c_uint64(seed), C.c_uint32(epoch))
    if rc != 0:
        raise ValueError(f"hns_oracle_reset failed: {rc}")


def reset_tasks(cfg, arrs, mask, seed, epoch, tasks, task_first):
    """`tasks` rows of the masked envs below `task_first` are OUTPUTS (the placement as sampled, include/hns.h): written in place."""
    b = as_struct(arrs)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    t = f32(tasks)
    rc = lib().hns_oracle_reset_tasks(C.byref(cfg), C.byref(b), _p(m), C.c_uint64(seed), C.c_uint32(epoch), _p(t), int(task_first))
    if rc != 0:
        raise ValueError(f"hns_oracle_reset_tasks failed: {rc}")
    if t is not tasks:
        np.copyto(tasks, t.reshape(np.shape(tasks)))


def quat_rotate(q, v, inverse=False):
    q, v = f32(q).reshape(-1, 4), f32(v).reshape(-1, 3)
    out = np.empty_like(v)
    lib().hns_oracle_quat_rotate(len(q), _p(q), _p(v), _p(out), int(inverse))
    return out


def euler_to_quat(rpy):
    rpy = f32(rpy).reshape(-1, 3)
    q = np.empty((len(rpy), 4), np.float32)
    lib().hns_oracle_euler_to_quat(len(rpy), _p(rpy), _p(q))
    return q


def elementary(x):
    x = f32(x).ravel()
    e, t, s, c = (np.empty_like(x) for _ in range(4))
    lib().hns_oracle_elementary(len(x), _p(x), _p(e), _p(t), _p(s), _p(c))
    return e, t, s, c


def rotor(cfg, cmd, throttle):
    cmd = f32(cmd).reshape(-1, 4)
    throttle = f32(throttle).reshape(-1, 4).copy()
    thrust, moment = np.empty_like(cmd), np.empty_like(cmd)
    td = np.empty(len(cmd), np.float32)
    lib().hns_oracle_rotor(C.byref(cfg), len(cmd), _p(cmd), _p(throttle), _p(thrust), _p(moment), _p(td))
    return throttle, thrust, moment, td


def ctbr_pid(cfg, action, rot, angvel, reset_mask, prev_action, integ, last):
    action, rot, angvel = f32(action).reshape(-1, 4), f32(rot).reshape(-1, 4), f32(angvel).reshape(-1, 3)
    n = len(action)
    prev_action, integ, last = f32(prev_action).reshape(n, 4).copy(), f32(integ).reshape(n, 3).copy(), f32(last).reshape(n, 3).copy()
    m = None if reset_mask is None else np.ascontiguousarray(reset_mask, np.uint8).reshape(n)
    cmd, ctbr = np.empty((n, 4), np.float32), np.empty((n, 4), np.float32)
    aerr, tr = np.empty(n, np.float32), np.empty((n, 3), np.float32)
    lib().hns_oracle_ctbr_pid(C.byref(cfg), n, _p(action), _p(rot), _p(angvel), _p(m), _p(prev_action), _p(integ),
                              _p(last), _p(cmd), _p(aerr), _p(ctbr), _p(tr))
    return dict(cmd=cmd, ctbr=ctbr, aerr=aerr, target_rate=tr, prev_action=prev_action, integ=integ, last=last)


def downwash(pos, rot, tsum):
    pos, rot, tsum = f32(pos), f32(rot), f32(tsum)
    E, A = pos.shape[:2]
    f = np.empty_like(pos)
    lib().hns_oracle_downwash(E, A, _p(pos), _p(rot), _p(tsum), _p(f))
    return f


def blocked(cfg, dpos, tpos, cyl):
    dpos, tpos, cyl = f32(dpos), f32(tpos).reshape(len(dpos), 3), f32(cyl)
    E, A = dpos.shape[:2]
    out = np.empty((E, A), np.uint8)
    lib().hns_oracle_blocked(C.byref(cfg), E, A, cyl.shape[1], _p(dpos), _p(tpos), _p(cyl), _p(out))
    return out.astype(bool)


def prey(cfg, dpos, tpos, cyl, out_of_arena=None):
    dpos, tpos, cyl = f32(dpos), f32(tpos).reshape(len(dpos), 3), f32(cyl)
    E, A = dpos.shape[:2]
    force, vel = np.empty((E, 3), np.float32), np.empty((E, 3), np.float32)
    ooa = np.zeros(E, np.float32) if out_of_arena is None else f32(out_of_arena).reshape(E).copy()
    lib().hns_oracle_prey(C.byref(cfg), E, A, cyl.shape[1], _p(dpos), _p(tpos), _p(cyl), _p(force), _p(vel), _p(ooa))
    return force, vel, ooa


def integrate(cfg, ds, force_w, torque_b):
    ds = f32(ds).reshape(-1, 13).copy()
    fw, tb = f32(force_w).reshape(-1, 3), f32(torque_b).reshape(-1, 3)
    lib().hns_oracle_integrate(C.byref(cfg), len(ds), _p(ds), _p(fw), _p(tb))
    return ds


def obs_reward(cfg, arrs, thr_diff=None, do_reward=False):
    E, A, K = cfg.num_envs, cfg.num_agents, cfg.obs_max_cylinder
    b = as_struct(arrs)
    blocked_, bdet, knn = np.empty((E, A), np.uint8), np.empty(E, np.uint8), np.empty((E, A, K), np.uint8)
    td = f32(thr_diff if thr_diff is not None else np.zeros((E, A)))
    lib().hns_oracle_obs_reward(C.byref(cfg), C.byref(b), _p(td), int(do_reward), _p(blocked_), _p(bdet), _p(knn))
    return blocked_.astype(bool), bdet.astype(bool), knn.astype(bool)


def cell(cfg, x):
    return lib().hns_oracle_cell(C.byref(cfg), C.c_float(x))


def philox(k0, k1, c0, c1, c2, c3):
    out = (C.c_uint32 * 4)()
    lib().hns_oracle_philox(C.c_uint32(k0), C.c_uint32(k1), C.c_uint32(c0), C.c_uint32(c1), C.c_uint32(c2), C.c_uint32(c3), out)
    return list(out)


# ---- Hover (BASELINE config 1) --------------------------------------------------------------------
def alloc_hover_buffers(cfg):
    return {k: np.zeros(shape, dtype=dt) for k, (shape, dt) in abi.hover_buffer_shapes(cfg.num_envs).items()}


def _hover_struct(arrs):
    b = abi.HnsHoverBuffers()
    for k in abi.HOVER_BUFFER_FIELDS:
        assert arrs[k].flags["C_CONTIGUOUS"]
        setattr(b, k, arrs[k].ctypes.data)
    return b


def hover_step(cfg, hcfg, arrs, action):
    action = f32(action)
    b = _hover_struct(arrs)
    rc = lib().hns_oracle_hover_step(C.byref(cfg), C.byref(hcfg), C.byref(b), _p(action))
    assert rc == 0, rc


def hover_reset(cfg, hcfg, arrs, mask, seed, epoch):
    b = _hover_struct(arrs)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    rc = lib().hns_oracle_hover_reset(C.byref(cfg), C.byref(hcfg), C.byref(b), _p(m), C.c_uint64(seed), C.c_uint32(epoch))
    assert rc == 0, rc


def raycast(cfg, arrs, num_rays, max_range):
    b = as_struct(arrs)
    out = np.empty((cfg.num_envs, cfg.num_agents, num_rays), np.float32)
    lib().hns_oracle_raycast(C.byref(cfg), C.byref(b), int(num_rays), C.c_float(max_range), _p(out))
    return out


def alloc_tp_buffers(cfg, T=10, F=5):
    """Zero-initialised host arrays for every hns_tp_buffers field (weights included)."""
    return {k: np.zeros(shape if k != "packed" else (16,), dtype=dt)       # `packed` is the HIP kernel's scratch
            for k, (shape, dt) in abi.tp_buffer_shapes(cfg.num_envs, cfg.num_agents, T, F,
                                                       abi.tp_frame_dim(cfg.num_agents, cfg.num_cylinders, cfg.tp_use_obstacles)).items()}


def tp_observe(cfg, arrs, tp_arrs, fill, with_state=True):
    """hns_tp_observe on host arrays: frame append / window shift, TP_net forward, 20+3F-value rows."""
    b = as_struct(arrs)
    t = abi.HnsTpBuffers()
    for k in abi.TP_BUFFER_FIELDS:
        a = tp_arrs[k]
        assert a.flags["C_CONTIGUOUS"] and a.dtype in (np.float32, np.uint8)
        setattr(t, k, a.ctypes.data)
    if not with_state:
        t.state_drones = None
    T, F = tp_arrs["history"].shape[1], tp_arrs["pred"].shape[1]
    rc = lib().hns_oracle_tp_observe(C.byref(cfg), C.byref(b), C.byref(t), int(T), int(F), int(bool(fill)))
    assert rc == 0, rc


def fps(points, k, start=0):
    """Farthest-point sampling indices (int32 [k]) of points [n,d]."""
    pts = f32(points)
    out = np.empty(int(k), np.int32)
    rc = lib().hns_oracle_fps(_p(pts), int(pts.shape[0]), int(pts.shape[1]), int(k), int(start), _p(out))
    assert rc == 0, rc
    return out


def perturb_tasks(cfg, history, n_tasks, expand_cylinders, expand_step, seed):
    hist = f32(history)
    out = np.zeros((int(n_tasks), hist.shape[1]), np.float32)
    rc = lib().hns_oracle_perturb_tasks(C.byref(cfg), _p(hist), int(hist.shape[0]), _p(out), int(n_tasks), int(bool(expand_cylinders)),
                                        C.c_float(expand_step), C.c_uint64(seed))
    assert rc == 0, rc
    return out


def tasks_sane(cfg, tasks):
    t = f32(tasks)
    out = np.zeros(t.shape[0], np.uint8)
    lib().hns_oracle_tasks_sane(C.byref(cfg), _p(t), int(t.shape[0]), _p(out))
    return out.astype(bool)
