"""Property tests (hypothesis) of invariants the reference implies: they hold for the
oracle on arbitrary plausible states, and — because HIP == oracle bit-for-bit — for the kernels."""
import numpy as np
from hypothesis import given, settings, strategies as st

import hns_oracle as O
from hns_amd import abi, config


def _env(E, A, C, seed, max_len=30, **task):
    cfg = config.make_cfg({"num_agents": A, "cylinder": {"max_num": C, "min_num": min(4, C)},
                           "env": {"num_envs": E, "max_episode_length": max_len}, **task})
    c = config.resolve_hns_cfg(cfg)
    arrs = O.alloc_buffers(c)
    O.reset(c, arrs, None, seed, 0)
    return c, arrs


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), A=st.sampled_from([1, 2, 3, 4, 6]), C=st.sampled_from([3, 5, 8, 16]),
       scale=st.floats(0.1, 3.0))
def test_step_invariants(seed, A, C, scale):
    E = 24
    c, arrs = _env(E, A, C, seed)
    rng = np.random.default_rng(seed)
    for t in range(12):
        O.step(c, arrs, (rng.standard_normal((E, A, 4)) * scale).astype(np.float32))
    ds = arrs["drone_state"]
    np.testing.assert_allclose(np.linalg.norm(ds[..., 3:7].astype(np.float64), axis=-1), 1.0, atol=3e-6)   # unit quaternions
    assert (np.linalg.norm(ds[..., 7:10].astype(np.float64), axis=-1) <= 1.0).all()                        # max_linear_velocity
    assert (ds[..., 2] >= 0).all() and np.isfinite(ds).all()
    assert ((arrs["throttle"] >= 0) & (arrs["throttle"] <= 1)).all()
    assert (np.abs(arrs["pid_integ"][..., :3]) <= np.array([33.3, 33.3, 166.7], np.float32) + 1e-4).all()   # iLimit
    # observation structure (hideandseek.py:856-863): t x4, quaternion copy, heading/up unit vectors
    o = arrs["obs_self"]
    assert (o[..., 16:20] == (arrs["progress"] / np.float32(30))[:, None, None]).all() or np.allclose(o[..., 16], arrs["progress"][:, None] / 30, atol=1e-7)
    assert (o[..., 3:10] == ds[..., 3:10]).all()
    np.testing.assert_allclose(np.linalg.norm(o[..., 10:13], axis=-1), 1.0, atol=1e-5)
    np.testing.assert_allclose((o[..., 10:13] * o[..., 13:16]).sum(-1), 0.0, atol=1e-5)                     # heading _|_ up
    # cylinder rows: masked rows are all -5, others carry [rpos, height, size] of an ACTIVE cylinder (:759-778)
    oc = arrs["obs_cylinders"]
    masked = (oc == -5).all(-1)
    assert ((oc[..., 3] == np.float32(1.2)) & (oc[..., 4] == np.float32(0.1)))[~masked].all()
    n_active = (arrs["cylinders"][..., 2] > 0).sum(1)
    assert ((~masked).sum(-1) == np.minimum(n_active, 3)[:, None]).all()
    # k-nearest rows are sorted by distance among the active ones
    d = np.linalg.norm(oc[..., :3], axis=-1)
    d[masked] = np.inf
    assert (np.diff(d, axis=-1) >= -1e-6).all()
    # evader speed quirk: every axis is +-v_prey up to the 1e-5 regulariser (:741)
    assert (np.abs(arrs["target_vel"]) <= 1.3 + 1e-6).all()
    # stats: binary flags stay binary, counters monotone-bounded
    sidx = abi.STAT_NAMES.index
    assert set(np.unique(arrs["stats"][sidx("success")])) <= {0.0, 1.0}
    assert set(np.unique(arrs["stats"][sidx("out_of_arena")])) <= {0.0, 1.0}
    assert (arrs["stats"][sidx("first_capture_step")] <= 30).all() and (arrs["stats"][sidx("sum_detect_step")] <= 12).all()
    assert not arrs["done"].any()


@settings(max_examples=10, deadline=None)
@given(seed=st.integers(0, 2**31 - 1))
def test_done_and_stats_division(seed):
    E, A, L = 16, 3, 9
    c, arrs = _env(E, A, 5, seed, max_len=L, reset_extra_step=0)     # (with the extra physics step a reset moves every env: next test)
    rng = np.random.default_rng(seed)
    sidx = abi.STAT_NAMES.index
    for t in range(L):
        before = arrs["stats"].copy()
        O.step(c, arrs, rng.standard_normal((E, A, 4)).astype(np.float32))
        assert bool(arrs["done"].all()) == (t == L - 1)
    # on the done step the accumulated per-step means were divided by the episode length (:1017-1056)
    acc = before[sidx("collision_wall")] + 0.0
    assert (arrs["stats"][sidx("collision_wall")] <= (acc + 2.0) / L + 1e-6).all()
    assert (arrs["progress"] == L).all()
    # reset of a subset leaves the others untouched, except first_capture_step (:712, all envs)
    mask = np.zeros(E, np.uint8)
    mask[::2] = 1
    keep = {k: v.copy() for k, v in arrs.items()}
    O.reset(c, arrs, mask, seed, 1)
    for k in ("drone_state", "throttle", "target_pos", "cylinders", "progress", "obs_self"):
        assert (arrs[k][1::2] == keep[k][1::2]).all(), k
        assert not (arrs[k][::2] == keep[k][::2]).all(), k
    assert (arrs["stats"][sidx("first_capture_step")] == L).all()
    assert (arrs["progress"][::2] == 0).all() and not arrs["stats"][:, ::2][np.arange(24) != sidx("first_capture_step")].any()


@settings(max_examples=5, deadline=None)
@given(seed=st.integers(0, 2**31 - 1))
def test_reset_extra_step_moves_the_whole_scene(seed):
    """hideandseek.py:722-723 (task.reset_extra_step: 1): `_reset_idx` ends with one physics step of every env — no rotor forces, so a
    drone's new state is the integrator applied with zero force and torque, an evader moves by dt times the velocity it holds; the
    controller state is not touched (task.pid_reset: reference), the done flag of the reset envs is cleared."""
    E, A, L = 12, 3, 6
    c0, arrs0 = _env(E, A, 5, seed, max_len=L, reset_extra_step=0)
    c1, arrs1 = _env(E, A, 5, seed, max_len=L, reset_extra_step=1)
    assert c1.reset_extra_step == 1 and c1.pid_reset_on_reset == 0
    # first reset (all envs): the placement of the plain reset, then one free-fall step
    ds = O.integrate(c0, arrs0["drone_state"], np.zeros((E * A, 3), np.float32), np.zeros((E * A, 3), np.float32))
    assert (arrs1["drone_state"].reshape(E * A, 13) == ds).all() and (ds[:, 9] < 0).all()      # falling
    assert (arrs1["target_pos"] == arrs0["target_pos"]).all()                                    # the evader held no velocity yet
    rng = np.random.default_rng(seed)
    for t in range(L):
        O.step(c1, arrs1, rng.standard_normal((E, A, 4)).astype(np.float32))
    assert arrs1["done"].all()
    keep = {k: v.copy() for k, v in arrs1.items()}
    mask = np.zeros(E, np.uint8)
    mask[::3] = 1
    O.reset(c1, arrs1, mask, seed, 1)
    others = mask == 0
    n = int(others.sum()) * A
    ds = O.integrate(c1, keep["drone_state"][others], np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32))
    assert (arrs1["drone_state"][others].reshape(-1, 13) == ds).all()
    assert (arrs1["target_pos"][others] == keep["target_pos"][others] + keep["target_vel"][others] * np.float32(c1.dt)).all()
    assert (arrs1["progress"][others] == L).all() and arrs1["done"][others].all() and not arrs1["done"][mask == 1].any()
    assert (arrs1["pid_integ"] == keep["pid_integ"]).all() and (arrs1["pid_last_rate"][..., :3] == keep["pid_last_rate"][..., :3]).all()
    assert (arrs1["obs_self"][others][..., 3:10] == arrs1["drone_state"][others][..., 3:10]).all()       # the observation follows the moved state
    assert (arrs1["target_pos"][mask == 1] != keep["target_pos"][mask == 1]).any()


def test_reset_distribution_matches_reference_ranges():
    """hideandseek.py:283-313,576-607: sampling boxes, 9x9 grid cells, distinct free cells, active counts."""
    c, arrs = _env(4096, 3, 5, 1234, reset_extra_step=0)
    r = 0.9 / np.sqrt(2.0)
    p = arrs["drone_state"][..., :3]
    assert (p[..., 0] >= 0.1).all() and (p[..., 0] <= r - 0.1 + 1e-6).all() and (np.abs(p[..., 1]) <= r - 0.1 + 1e-6).all()
    assert (p[..., 2] >= 0.5).all() and (p[..., 2] <= 0.7).all()
    tp = arrs["target_pos"]
    assert (tp[:, 0] <= -0.1).all() and (tp[:, 0] >= -r + 0.1 - 1e-6).all()
    cyl = arrs["cylinders"]
    cells = np.rint(cyl[..., :2] / 0.2).astype(int) + 4
    assert ((cells >= 0) & (cells <= 8)).all()
    assert (np.sqrt(((cells - 4) ** 2).sum(-1)) < 4).all()                      # inside the disc
    flat = cells[..., 0] * 9 + cells[..., 1]
    assert all(len(set(row)) == 5 for row in flat)                                # distinct cells
    dcell = np.clip(np.rint(p[..., :2] / 0.2).astype(int) + 4, 0, 8)
    occupied = dcell[..., 0] * 9 + dcell[..., 1]
    assert not (flat[:, :, None] == occupied[:, None, :]).any()                   # never on a drone's cell
    n_active = (cyl[..., 2] > 0).sum(1)
    assert set(np.unique(n_active)) == {4, 5}                                     # randint[min_num, max_num]
    assert abs((n_active == 4).mean() - 0.5) < 0.05
    assert ((cyl[..., 2] == np.float32(0.6)) | (cyl[..., 2] == -20)).all()
    # uniformity of the chosen cells over the 45 free cells (chi-square-ish bound)
    counts = np.bincount(flat.ravel(), minlength=81)
    used = counts[counts > 0]
    assert len(used) == 45 - 0 or len(used) >= 40
    assert used.std() / used.mean() < 0.25


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), A=st.sampled_from([1, 3, 6]), C=st.sampled_from([3, 8, 16]))
def test_line_of_sight_column_describes_the_state_it_sits_in(seed, A, C):
    """include/hns.h: pid_last_rate[..., 3] = the pursuer's line-of-sight flag in the state the buffers hold.  After resets
    (full and masked) and steps it equals a fresh evaluation of the exported positions — the property the step kernel's
    carry-over rests on; the GPU twin of this test is in test_hip_parity.py."""
    E = 40
    c, arrs = _env(E, A, C, seed, max_len=7)
    rng = np.random.default_rng(seed)
    epoch = 1
    for t in range(16):
        fresh = O.blocked(c, arrs["drone_state"][..., :3], arrs["target_pos"], arrs["cylinders"])
        assert np.array_equal(arrs["pid_last_rate"][..., 3], fresh.astype(np.float32)), t
        O.step(c, arrs, rng.standard_normal((E, A, 4)).astype(np.float32))
        if arrs["done"].any():
            mask = arrs["done"].copy()
            mask[::4] = 0                                     # some done envs keep running: flags of both kinds side by side
            O.reset(c, arrs, mask, seed, epoch)
            epoch += 1
