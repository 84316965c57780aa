"""Adaptive Environment Generator (BASELINE config 4): host-side GenBuffer logic
(reference hideandseek_envgen.py:209-377) and the task-vector reset."""
import numpy as np
import pytest
import torch

import hns_oracle as O
from hns_amd import config
from hns_amd.envgen import GenBuffer, farthest_point_sampling


def test_farthest_point_sampling_properties():
    g = torch.Generator().manual_seed(0)
    pts = torch.rand(400, 5, generator=g)
    idx = farthest_point_sampling(pts, 50, start=3)
    assert idx[0] == 3 and len(set(idx.tolist())) == 50
    # greedy property: every new point is the farthest from the already chosen set
    chosen = [3]
    for i in range(1, 50):
        d = torch.cdist(pts, pts[chosen]).min(1).values
        assert abs(float(d[idx[i]]) - float(d.max())) < 1e-6
        chosen.append(int(idx[i]))
    # the sample covers the cloud far better than a prefix
    cover_fps = torch.cdist(pts, pts[idx]).min(1).values.max()
    cover_prefix = torch.cdist(pts, pts[:50]).min(1).values.max()
    assert cover_fps < cover_prefix
    assert farthest_point_sampling(pts[:10], 50).tolist() == list(range(10))


def _valid_tasks(gb, n, rng):
    """n tasks with all objects on distinct free grid cells."""
    free = np.argwhere(gb.grid_map == 0)
    A, Cn = gb.num_agents, gb.num_cylinders
    out = []
    for _ in range(n):
        NM = A + getattr(gb, "num_targets", 1)
        cells = free[rng.permutation(len(free))[:NM + Cn]]
        xy = (cells - gb.num_grid // 2) * gb.grid_size
        z = np.concatenate([np.full(NM, 1.2), np.where(np.arange(Cn) < 3, 0.6, -20.0)])
        out.append(np.concatenate([xy, z[:, None]], axis=1).reshape(-1))
    return np.asarray(out, dtype=np.float32)


def test_genbuffer_samplenearby_and_history():
    rng = np.random.default_rng(1)
    gb = GenBuffer(3, 5, seed=2, buffer_length=64)
    assert gb.task_dim == 18 + 3 * 3 and int((gb.grid_map == 0).sum()) == 45   # reference task_dim, 45 free cells
    base = _valid_tasks(gb, 40, rng)
    assert gb.sanity_ok(base).all()
    bad = base.copy()
    bad[:, 3:5] = bad[:, 0:2]                       # two drones in one cell -> rejected (:187-207)
    assert not gb.sanity_ok(bad).any()
    gb.init_history(base)
    near = gb.samplenearby(500, expand_cylinders=1, expand_step=0.1)
    assert near.shape == (500, gb.task_dim) and gb.sanity_ok(near).all()
    b = gb.task_bounds()
    assert (near >= b[:, 0] - 1e-6).all() and (near <= b[:, 1] + 1e-6).all()
    assert (np.abs(near[:, :12].reshape(500, 4, 3)[..., 2] - 1.2) <= 0.1 + 1e-6).all()   # reference z window
    # weights over eval_iter episodes -> mean; R_min..R_max filter; FPS trim to buffer_length
    gb.insert(near)
    for _ in range(3):
        gb.insert_weights(rng.random(500) > 0.4)
    gb.update()
    assert gb._weight_buffer.shape == (500, 1) and gb._state_buffer.shape == near.shape
    keep = ((gb._weight_buffer <= 0.9) & (gb._weight_buffer >= 0.5)).reshape(-1)
    gb.insert_history(gb._state_buffer[keep])
    assert gb._history_buffer.shape == (64, gb.task_dim)
    easy = GenBuffer(3, 5, seed=0, buffer_length=20).init_easy_cases()
    assert easy.shape == (20, 4, 3) and (np.abs(easy[..., 2] - 0.6) <= 0.1).all()


def test_easy_cases_take_the_cells_a_flood_reaches_first():
    """init_easy_cases ranks cells by a precomputed flood table (up to four pursuers) or floods literally (more); here the reference's
    queue flood as written, hideandseek_envgen.py:246-262, including its `if len(found) == 4: break` (the fourth cell found is not
    enqueued, the rest of that expansion is skipped)."""
    from collections import deque
    for A, seed in ((3, 1), (4, 2), (6, 3), (7, 4)):
        gb = GenBuffer(A, 5, seed=seed, buffer_length=40)
        easy = gb.init_easy_cases()
        n, free = gb.num_grid, np.argwhere(gb.grid_map == 0)
        rng = np.random.default_rng(seed)
        for k in range(40):
            sx, sy = free[rng.integers(len(free))]
            seen = {(sx, sy)}
            todo, got = deque([(sx, sy)]), []
            while todo and len(got) < A:
                cx, cy = todo.popleft()
                for dx, dy in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                    q = (cx + dx, cy + dy)
                    if 0 <= q[0] < n and 0 <= q[1] < n and q not in seen:
                        seen.add(q)
                        if gb.grid_map[q] == 0:
                            got.append(q)
                            if len(got) == 4:
                                break
                        todo.append(q)
            want = np.clip((np.array(got[:A] + [(sx, sy)], dtype=np.float64) - n // 2) * gb.grid_size, -gb.boundary, gb.boundary)
            np.testing.assert_array_equal(easy[k, :, :2], want.astype(np.float32))


def test_oracle_reset_from_task_vectors():
    E, A, Cn = 40, 3, 5
    cfg = config.make_cfg({"num_agents": A, "reset_extra_step": 0, "cylinder": {"max_num": Cn, "min_num": 4}, "env": {"num_envs": E}})   # the placement itself
    c = config.resolve_hns_cfg(cfg)
    gb = GenBuffer(A, Cn, seed=3)
    tasks = _valid_tasks(gb, E, np.random.default_rng(4))
    arrs = O.alloc_buffers(c)
    O.reset_tasks(c, arrs, None, 5, 0, tasks, 16)
    t3 = tasks.reshape(E, A + 1 + Cn, 3)
    np.testing.assert_array_equal(arrs["drone_state"][16:, :, :3], t3[16:, :A])
    np.testing.assert_array_equal(arrs["target_pos"][16:], t3[16:, A])
    np.testing.assert_array_equal(arrs["cylinders"][16:], t3[16:, A + 1:])
    # envs below task_first are sampled as in the plain reset (same Philox draws)
    ref = O.alloc_buffers(c)
    O.reset(c, ref, None, 5, 0)
    np.testing.assert_array_equal(arrs["drone_state"][:16], ref["drone_state"][:16])
    np.testing.assert_array_equal(arrs["cylinders"][:16], ref["cylinders"][:16])
    np.testing.assert_allclose(np.linalg.norm(arrs["drone_state"][..., 3:7], axis=-1), 1.0, atol=1e-6)
    assert np.isfinite(arrs["obs_self"]).all() and not arrs["progress"].any()


def test_oracle_task_reset_archives_the_placement_as_sampled():
    """with the default task.reset_extra_step = 1 the rows of the uniformly sampled envs (below task_first) are written
    with the placement AS SAMPLED — what the reference inserts into the generator (`tasks_unif`, hideandseek_envgen.py:883-895; the scene's
    sim.step comes afterwards, :1013) — not with the state after that step: drones would sit g dt^2 lower, the evader displaced by v dt."""
    E, A, Cn, first = 48, 3, 5, 20
    gb = GenBuffer(A, Cn, seed=3)
    base = _valid_tasks(gb, E, np.random.default_rng(4))
    out = {}
    for extra in (0, 1):
        c = config.resolve_hns_cfg(config.make_cfg({"num_agents": A, "reset_extra_step": extra, "cylinder": {"max_num": Cn, "min_num": 4}, "env": {"num_envs": E}}))
        arrs = O.alloc_buffers(c)
        arrs["target_vel"][:] = np.random.default_rng(5).uniform(-1.3, 1.3, arrs["target_vel"].shape)   # a held evader velocity: the extra step moves the evader
        tasks = base.copy()
        mask = np.ones(E, np.uint8)
        mask[[3, 30]] = 0                                             # two envs are not reset: their rows stay what they were
        O.reset_tasks(c, arrs, mask, 5, 0, tasks, first)
        out[extra] = (tasks, arrs)
    t0, a0 = out[0]
    t1, a1 = out[1]
    np.testing.assert_array_equal(t1, t0)                             # the archive does not depend on the extra step
    np.testing.assert_array_equal(t1[first:], base[first:])           # given tasks: untouched
    np.testing.assert_array_equal(t1[3], base[3])                     # not masked: untouched
    placed0 = np.concatenate([a0["drone_state"][..., :3].reshape(E, -1), a0["target_pos"].reshape(E, -1), a0["cylinders"].reshape(E, -1)], axis=1)
    keep = np.ones(E, bool)
    keep[[3, 30]] = False
    np.testing.assert_array_equal(t1[keep], placed0[keep])            # = where the bodies sit when nothing has been stepped
    placed1 = a1["drone_state"][..., :3].reshape(E, -1)
    dz = (t1[:first, 2:3 * A:3] - placed1[:first, 2::3])[keep[:first]]
    assert (dz > 5e-4).all() and (dz < 2e-3).all()                    # ... while the stepped drones are ~g dt^2 = 0.98 mm lower
    assert np.abs(a1["target_pos"] - t1[:, 3 * A:3 * A + 3])[keep].max() > 1e-3


# (The tests that check "every env sits ON its task vector after a reset" run with task.reset_extra_step = 0: the placement itself, without the one
#  physics step the reference appends to `_reset_idx`, hideandseek_envgen.py:1012-1013; that step is pinned by tests/test_reset_pid.py and
#  test_oracle_properties.py, and the other generator tests here run with it, as the default has it.)
@pytest.mark.gpu
def test_envgen_env_on_gpu():
    from hns_amd.env import HideAndSeek
    from hns_amd.envgen import HideAndSeek_envgen
    E, L = 512, 6
    cfg = config.make_cfg({"name": "HideAndSeek_envgen", "reset_extra_step": 0, "num_agents": 3, "ratio_unif": 0.3, "eval_iter": 2, "R_min": 0.0, "R_max": 1.0,
                           "use_particle_generator": 1, "expand_cylinders": 1,
                           "cylinder": {"max_num": 5, "min_num": 4}, "env": {"num_envs": E, "max_episode_length": L}})
    env = HideAndSeek.REGISTRY[cfg.task.name](cfg, headless=True)
    assert isinstance(env, HideAndSeek_envgen)
    env.set_seed(3)
    env.reset()
    assert env.num_unif == E and env.all_tasks.shape == (E, 27)      # empty history: everything uniform
    first_tasks = env.all_tasks.copy()
    g = torch.Generator(device=env.device).manual_seed(0)
    for ep in range(5):
        for t in range(L):
            td = env.step(env.rand_step_input(torch.randn(E, 3, 4, generator=g, device=env.device)))
        assert bool(td[("next", "done")].all())
        rtd = env.rand_step_input()
        rtd.set("_reset", td[("next", "done")].squeeze(-1))
        epoch = env.reset_epoch
        env.reset(rtd)
        st = env.export_state()
        placed = np.concatenate([st["drone_state"][..., :3].reshape(E, -1), st["target_pos"], st["cylinders"].reshape(E, -1)], axis=1)
        np.testing.assert_array_equal(placed, env.all_tasks)          # every env sits on its task vector
        if ep == 0:
            np.testing.assert_array_equal(env.all_tasks, first_tasks)  # replayed for eval_iter episodes
        # HIP task reset == oracle task reset
        host = O.alloc_buffers(env.hcfg)
        O.reset_tasks(env.hcfg, host, None, env.seed, epoch, env.all_tasks, 0 if ep % 2 == 0 else env.num_unif)
        if ep % 2 == 0:
            for k in ("drone_state", "target_pos", "cylinders", "obs_self", "obs_others", "obs_cylinders", "throttle"):
                np.testing.assert_array_equal(host[k], st[k], err_msg=k)
    hist = len(env.gen_buffer._history_buffer)
    assert hist > 0 and float(env.stats["history_buffer"][0]) == hist
    assert env.num_unif == E - min(hist, int(E * 0.7)) or env.update_iter != 0
    assert float(env.stats["ratio_unif"][0]) == pytest.approx(0.3)
    assert sum(float(env.stats[f"ratio_cylinders_{i}"][0]) for i in range(6)) == pytest.approx(1.0)
    assert env.generator_seconds > 0


@pytest.mark.gpu
def test_envgen_history_trim_on_gpu():
    """More kept tasks than the buffer holds: the history is trimmed to 5000 by hns_fps, every kept row
    is one of the inserted tasks, and the next batch is drawn around them by hns_perturb_tasks."""
    from hns_amd.envgen import HideAndSeek_envgen, GenBuffer
    E, L = 8192, 3
    cfg = config.make_cfg({"name": "HideAndSeek_envgen", "num_agents": 3, "ratio_unif": 0.3, "eval_iter": 1, "R_min": 0.0, "R_max": 1.0,
                           "use_particle_generator": 1, "cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E, "max_episode_length": L}})
    env = HideAndSeek_envgen(cfg)
    env.set_seed(1)
    env.reset()
    batch0 = env.all_tasks.copy()
    for ep in range(2):
        for t in range(L):
            td = env.step(env.rand_step_input())
        rtd = env.rand_step_input()
        rtd.set("_reset", td[("next", "done")].squeeze(-1))
        env.reset(rtd)
        hist = env.gen_buffer._history_buffer
        assert hist.shape == (5000, 36) and len(np.unique(hist, axis=0)) == 5000
        if ep == 0:
            rows = {r.tobytes() for r in batch0}
            assert all(r.tobytes() in rows for r in hist)                 # FPS selects, it does not alter
            # the trim equals the oracle's FPS on the same normalised points with the same start index
            lo, hi = batch0.min(0), batch0.max(0)
            normed = ((batch0 - lo) / (hi - lo + np.float32(1e-5))).astype(np.float32)
            start = int(np.flatnonzero((batch0 == hist[0]).all(1))[0])
            ref = O.fps(normed, 5000, start)
            np.testing.assert_array_equal(hist, batch0[ref])
    assert env.num_unif == E - min(5000, int(E * 0.7))
    new = env.all_tasks[env.num_unif:]
    gb = GenBuffer(3, 8)
    assert gb.sanity_ok(new).mean() > 0.95                                # perturbed tasks pass the grid check (fallbacks may not)


@pytest.mark.gpu
def test_envgen_whole_env_at_65536_envs():
    """BASELINE config 4 at its stated size as a whole env (the farthest-point trim alone is covered up to 200 000 points in
    test_hip_envgen.py): three task batches, 70 536 -> 5000 trims, perturbed batches, every buffer finite."""
    from hns_amd.envgen import HideAndSeek_envgen, GenBuffer
    E, L = 65536, 3
    env = HideAndSeek_envgen(config.make_cfg({"name": "HideAndSeek_envgen", "num_agents": 3, "eval_iter": 1, "R_min": 0.0, "R_max": 1.0, "ratio_unif": 0.3,
                                              "use_particle_generator": 1, "cylinder": {"max_num": 8, "min_num": 8},
                                              "env": {"num_envs": E, "max_episode_length": L}}))
    env.set_seed(4)
    env.reset()
    rows = {r.tobytes() for r in env.all_tasks}
    for ep in range(3):
        for t in range(L):
            td = env.step(env.rand_step_input())
        assert bool(td[("next", "done")].all())
        rtd = env.rand_step_input()
        rtd.set("_reset", td[("next", "done")].squeeze(-1))
        env.reset(rtd)
        hist = env.gen_buffer._history_buffer
        assert hist.shape == (5000, 36) and len(np.unique(hist, axis=0)) == 5000
        if ep == 0:
            assert all(r.tobytes() in rows for r in hist)                 # the trim selects among the tasks that were run
        assert env.num_unif == E - 5000                                   # min(len(history), int(E 0.7)) perturbed tasks
        assert GenBuffer(3, 8).sanity_ok(env.all_tasks[env.num_unif:]).mean() > 0.95
    assert env.check_finite(deep=True) and float(env.stats["history_buffer"][0]) == 5000


@pytest.mark.gpu
def test_envgen_archives_sampled_tasks_with_the_extra_step_on_gpu():
    """The default path (task.reset_extra_step: 1) on the device: the generator's task batch holds the uniform tasks as sampled — equal to the
    batch of a twin env that runs without the extra step, and to the oracle's rows — while the bodies have moved one dt."""
    from hns_amd.envgen import HideAndSeek_envgen
    E, L = 320, 5
    envs = {}
    for extra in (0, 1):
        cfg = config.make_cfg({"name": "HideAndSeek_envgen", "reset_extra_step": extra, "num_agents": 3, "ratio_unif": 0.3, "eval_iter": 1, "R_min": 0.0, "R_max": 1.0,
                               "use_particle_generator": 1, "cylinder": {"max_num": 6, "min_num": 3}, "env": {"num_envs": E, "max_episode_length": L}})
        env = HideAndSeek_envgen(cfg)
        env.set_seed(11)
        env.reset()
        envs[extra] = env
    assert envs[1].hcfg.reset_extra_step == 1 and envs[1].num_unif == E
    np.testing.assert_array_equal(envs[1].all_tasks, envs[0].all_tasks)
    st0, st1 = envs[0].export_state(), envs[1].export_state()
    placed0 = np.concatenate([st0["drone_state"][..., :3].reshape(E, -1), st0["target_pos"], st0["cylinders"].reshape(E, -1)], axis=1)
    np.testing.assert_array_equal(envs[1].all_tasks, placed0)
    dz = envs[1].all_tasks[:, 2:9:3] - st1["drone_state"][..., 2]
    assert (dz > 5e-4).all() and (dz < 2e-3).all()
    host = O.alloc_buffers(envs[1].hcfg)
    tasks = np.zeros_like(envs[1].all_tasks)
    O.reset_tasks(envs[1].hcfg, host, None, envs[1].seed, 0, tasks, E)
    np.testing.assert_array_equal(tasks, envs[1].all_tasks)
    for k in ("drone_state", "target_pos", "cylinders", "obs_self"):
        np.testing.assert_array_equal(host[k], st1[k], err_msg=k)
    # second batch: history present -> [uniform | perturbed]; the uniform rows are again the sampled placement, the perturbed rows untouched
    env = envs[1]
    for t in range(L):
        td = env.step(env.rand_step_input())
    rtd = env.rand_step_input()
    rtd.set("_reset", td[("next", "done")].squeeze(-1))
    env.reset(rtd)
    assert 0 < env.num_unif < E
    st = env.export_state()
    placed = np.concatenate([st["drone_state"][..., :3].reshape(E, -1), st["target_pos"], st["cylinders"].reshape(E, -1)], axis=1)
    dz = env.all_tasks[:, 2:9:3] - st["drone_state"][..., 2]
    assert (dz > 5e-4).all() and (dz < 2e-3).all()                    # every row (sampled or perturbed) is where the drones were PLACED
    np.testing.assert_array_equal(env.all_tasks[:, 12:], placed[:, 12:])    # cylinders do not move
    moved = np.abs(env.all_tasks[:, 9:12] - placed[:, 9:12]).max(axis=1)    # the evader flew one dt with the velocity it held at the episode's end
    assert (moved > 1e-3).mean() > 0.9 and moved.max() < 1.3 * 0.01 * 1.01


@pytest.mark.gpu
def test_envgen_with_trajectory_predictor_on_gpu():
    """Both reference defaults together (use_particle_generator + use_TP_net): the reset's observation pass runs
    the predictor too, so the 35-value rows handed out by reset() belong to the NEW placement."""
    from hns_amd.envgen import HideAndSeek_envgen
    E, L = 256, 4
    cfg = config.make_cfg({"name": "HideAndSeek_envgen", "num_agents": 3, "eval_iter": 1, "R_min": 0.0, "R_max": 1.0,
                           "use_particle_generator": 1, "cylinder": {"max_num": 5, "min_num": 3}, "env": {"num_envs": E, "max_episode_length": L}},
                          algo={"use_TP_net": 1})
    env = HideAndSeek_envgen(cfg)
    env.set_seed(2)
    td = env.reset()
    for ep in range(2):
        for t in range(L):
            td = env.step(env.rand_step_input())
        rtd = env.rand_step_input()
        rtd.set("_reset", td[("next", "done")].squeeze(-1))
        td = env.reset(rtd)
        ss = td[("agents", "observation", "state_self")][:, :, 0]
        assert ss.shape == (E, 3, 35)
        b = env._bufs
        pred = env._tp_bufs["pred"]                                   # [E,5,3]
        rpos_pred = (b["drone_state"][..., None, :3] - pred[:, None]).reshape(E, 3, 15)
        assert torch.equal(ss[..., 3:18], rpos_pred)                  # rows were rebuilt on the reset state
        assert torch.equal(ss[..., 18:], b["obs_self"][..., 3:])
        assert torch.equal(td[("agents", "TP", "TP_input")][:, -1, 7:], b["drone_state"][..., :3].reshape(E, -1))


def test_grid_sanity_check_matches_reference(golden):
    """The reference's own `sanity_check` + `continuous_to_grid` (hideandseek_envgen.py:145-207, executed as they
    are) on 3000 task vectors, half of the bodies exactly on cell centres / edges: the host GenBuffer and the C
    oracle (whose perturbation kernel the HIP one matches bit for bit) agree on every one."""
    import hns_oracle as O
    g = golden("g_envgen_sanity")
    A, Cn, N = (int(x) for x in g["meta"])
    gb = GenBuffer(A, Cn)
    assert np.array_equal(gb.grid_map, g["disc"])
    assert np.array_equal(gb.sanity_ok(g["tasks"]), g["ok"])
    c = config.resolve_hns_cfg(config.make_cfg({"num_agents": A, "cylinder": {"max_num": Cn, "min_num": 2}, "env": {"num_envs": 8}}))
    assert np.array_equal(O.tasks_sane(c, g["tasks"]), g["ok"])
    assert 0.05 < g["ok"].mean() < 0.95


def test_two_evader_task_vectors():
    """Two-evader extension: task vectors [pursuers | evader 0 | evader 1 | cylinder slots] through the host GenBuffer, the oracle's sanity
    check, perturbation and task reset."""
    import hns_oracle as O
    E, A, Cn = 48, 4, 6
    c = config.resolve_hns_cfg(config.make_cfg({"num_agents": A, "num_targets": 2, "reset_extra_step": 0, "cylinder": {"max_num": Cn, "min_num": 3}, "env": {"num_envs": E}}))
    gb = GenBuffer(A, Cn, seed=1, num_targets=2, buffer_length=64)
    assert gb.task_dim == 3 * (A + 2 + Cn) and gb.task_bounds().shape == (gb.task_dim, 2)
    tasks = _valid_tasks(gb, E, np.random.default_rng(2))
    assert gb.sanity_ok(tasks).all() and O.tasks_sane(c, tasks).all()
    clash = tasks.copy()
    clash[:, 3 * (A + 1):3 * (A + 1) + 2] = clash[:, 3 * A:3 * A + 2]                  # the second evader on the first one's cell
    assert not gb.sanity_ok(clash).any() and not O.tasks_sane(c, clash).any()
    gb.init_history(tasks)
    near = gb.samplenearby(200, expand_cylinders=1, expand_step=0.1)
    assert near.shape == (200, gb.task_dim) and gb.sanity_ok(near).all()
    pert = O.perturb_tasks(c, tasks, 300, 1, 0.1, seed=3)
    assert pert.shape == (300, gb.task_dim) and gb.sanity_ok(pert).mean() > 0.9 and np.array_equal(gb.sanity_ok(pert), O.tasks_sane(c, pert).astype(bool))
    moved = np.abs(pert.reshape(300, -1, 3)[:, :A + 2] - 1.0).max() > 0                 # both evaders are jittered like pursuers
    assert moved
    arrs = O.alloc_buffers(c)
    O.reset_tasks(c, arrs, None, 5, 0, tasks, 0)
    t3 = tasks.reshape(E, A + 2 + Cn, 3)
    np.testing.assert_array_equal(arrs["drone_state"][:, :, :3], t3[:, :A])
    np.testing.assert_array_equal(arrs["target_pos"], t3[:, A:A + 2])
    np.testing.assert_array_equal(arrs["cylinders"], t3[:, A + 2:])
    with pytest.raises(NotImplementedError):
        gb.init_easy_cases()


# ---- device-side generator pieces: oracle restatements -----------------------------------
def test_oracle_fps_matches_torch_reference():
    """Integer coordinates: squared distances are exact in fp32 whatever the summation order, so the
    oracle's index sequence must equal the plain-torch FPS (ties -> lower index in both)."""
    import hns_oracle as O
    from hns_amd.envgen import farthest_point_sampling
    rng = np.random.default_rng(5)
    for n, d, k in ((50, 3, 49), (700, 36, 120), (2000, 7, 64)):
        pts = rng.integers(-20, 21, size=(n, d)).astype(np.float32)
        pts[n // 2] = pts[3]                                   # duplicates: zero distances and ties
        ref = farthest_point_sampling(torch.from_numpy(pts), k, start=7).numpy()
        got = O.fps(pts, k, start=7)
        assert (got == ref).all()
        assert len(set(got.tolist())) == k


def test_oracle_perturb_tasks_properties():
    import hns_oracle as O
    A, Cn = 3, 5
    cfg = config.make_cfg({"num_agents": A, "cylinder": {"max_num": Cn, "min_num": 2}, "env": {"num_envs": 64}})
    c = config.resolve_hns_cfg(cfg)
    gb = GenBuffer(A, Cn)
    # history = valid tasks: placements produced by the oracle's own reset
    arrs = O.alloc_buffers(c)
    O.reset(c, arrs, None, 3, 0)
    hist = np.concatenate([arrs["drone_state"][..., :3].reshape(64, -1), arrs["target_pos"], arrs["cylinders"].reshape(64, -1)], axis=1)
    hist[:, 2:3 * (A + 1):3] = 1.2                              # the reference's z window sits around max_height (:320-333)
    hist = hist[gb.sanity_ok(hist)]                             # uniform resets may put two pursuers into one cell
    assert len(hist) > 20
    out = O.perturb_tasks(c, hist, 500, 0, 0.1, seed=11)
    assert out.shape == (500, 3 * (A + 1 + Cn)) and gb.sanity_ok(out).all()
    b = gb.task_bounds()
    assert (out >= b[:, 0] - 1e-6).all() and (out <= b[:, 1] + 1e-6).all()
    # every task is a jittered copy of SOME history entry: cylinders untouched, bodies within expand_step
    cyl_match = (np.abs(out[:, None, 3 * (A + 1):] - hist[None, :, 3 * (A + 1):]).max(-1) == 0)
    body_close = (np.abs(out[:, None, :3 * (A + 1)] - hist[None, :, :3 * (A + 1)]).max(-1) <= 0.1 + 1e-6)
    assert (cyl_match & body_close).any(1).all()
    assert (np.abs(out[:, None, :] - hist[None]).max(-1).min(1) > 0).mean() > 0.9      # and most really moved
    assert np.array_equal(out, O.perturb_tasks(c, hist, 500, 0, 0.1, seed=11))
    assert not np.array_equal(out, O.perturb_tasks(c, hist, 500, 0, 0.1, seed=12))
    moved = O.perturb_tasks(c, hist, 500, 1, 0.1, seed=11)                               # cylinders hop by whole cells
    dxy = (moved[:, None, 3 * (A + 1):] - hist[None, :, 3 * (A + 1):]).reshape(500, len(hist), Cn, 3)[..., :2]
    cells = np.abs(dxy / 0.2 - np.rint(dxy / 0.2)).max((-1, -2)).min(1)
    assert (cells < 1e-4).all() and gb.sanity_ok(moved).all()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(12)))
def test_envgen_random_configurations_on_gpu(seed):
    """Seeded sweep over the generator's configuration space (pursuers, slots, one or two evaders, batch size, eval_iter, ratio_unif, cylinder
    expansion, easy initial cases): every env sits on its task vector after every reset, the envs placed from history tasks equal the oracle's
    task reset bit for bit, the history never exceeds its length, the statistics stay finite and consistent."""
    from hns_amd.envgen import HideAndSeek_envgen
    r = np.random.RandomState(900 + seed)
    for _ in range(50):
        A, Cn, NT = int(r.randint(1, 7)), int(r.randint(2, 11)), 2 if r.rand() < 0.35 else 1
        E = int(r.choice([192, 1000, 6144]))
        task = {"name": "HideAndSeek_envgen", "reset_extra_step": 0, "num_agents": A, "num_targets": NT, "ratio_unif": float(r.choice([0.0, 0.3, 0.7])), "eval_iter": int(r.randint(1, 4)),
                "R_min": float(r.choice([0.0, 0.2])), "R_max": 1.0, "use_particle_generator": 1, "expand_cylinders": int(r.rand() < 0.5),
                "expand_step": float(r.choice([0.05, 0.1])), "use_init_easy": int(NT == 1 and r.rand() < 0.4), "catch_radius": float(r.choice([0.3, 0.6])),
                "cylinder": {"max_num": Cn, "min_num": int(r.randint(0, Cn + 1)), "size": float(r.choice([0.1, 0.12]))},
                "env": {"num_envs": E, "max_episode_length": int(r.randint(3, 7))}}
        try:
            config.resolve_hns_cfg(config.make_cfg(task))
            break
        except ValueError:
            continue
    L = task["env"]["max_episode_length"]
    env = HideAndSeek_envgen(config.make_cfg(task))
    env.set_seed(seed)
    env.reset()
    assert env.all_tasks.shape == (E, 3 * A + 3 * NT + 3 * Cn)
    g = torch.Generator(device=env.device).manual_seed(seed)
    for ep in range(2 * task["eval_iter"] + 1):
        for t in range(L):
            td = env.step(env.rand_step_input(torch.randn(E, A, 4, generator=g, device=env.device) * 0.5))
        assert bool(td[("next", "done")].all())
        rtd = env.rand_step_input()
        rtd.set("_reset", td[("next", "done")].squeeze(-1))
        epoch = env.reset_epoch
        env.reset(rtd)
        st = env.export_state()
        placed = np.concatenate([st["drone_state"][..., :3].reshape(E, -1), st["target_pos"].reshape(E, -1), st["cylinders"].reshape(E, -1)], axis=1)
        np.testing.assert_array_equal(placed, env.all_tasks, err_msg=f"seed {seed} {task} episode {ep}")
        host = O.alloc_buffers(env.hcfg)
        O.reset_tasks(env.hcfg, host, None, env.seed, epoch, env.all_tasks, env.num_unif)
        for k in ("drone_state", "target_pos", "cylinders", "obs_self", "obs_others", "obs_cylinders", "throttle"):
            np.testing.assert_array_equal(host[k][env.num_unif:], st[k][env.num_unif:], err_msg=f"seed {seed} {task} episode {ep}: {k}")
        assert len(env.gen_buffer) <= env.gen_buffer.buffer_length and 0 <= env.num_unif <= E
        for k in ("success_buffer", "success_unif", "history_buffer", "ratio_unif"):
            # as in the reference, success_unif is the mean of an EMPTY slice (NaN) in an episode without uniform tasks (hideandseek_envgen.py:1241-1243)
            ok = torch.isfinite(env.stats[k]).all() or (k == "success_unif" and task["ratio_unif"] == 0.0 and torch.isnan(env.stats[k]).all())
            assert bool(ok), k
        assert float(env.stats["history_buffer"][0]) == len(env.gen_buffer)
    assert env.check_finite()                                  # the state buffers (the logging statistic above is not one of them)


# ---- A12 against the reference's own GenBuffer and curriculum statements (tests/golden/g_genbuffer.npz, make_golden.py::gen_genbuffer) --------------
def _scalar(g, key):
    v = np.asarray(g[key], dtype=np.float64)
    assert v.ndim == 0, key
    return float(v)


def _replay_batches(g, make_buffer, to_dev=lambda x: x):
    """Both task batches of the fixture through insert -> insert_weights x eval_iter -> curriculum_update on `make_buffer()`; yields what to compare."""
    from hns_amd.envgen import curriculum_update
    A, Cn, E, eval_iter, B = (int(x) for x in g["meta"])
    R_min, R_max = (float(x) for x in g["r_bounds"])
    gb = make_buffer(A, Cn, B)
    for batch in range(2):
        gb.insert(to_dev(g[f"b{batch}_tasks"]))
        for ep in range(eval_iter):
            gb.insert_weights(to_dev(g[f"b{batch}_success{ep}"]))
        gb.fps_start = int(g[f"b{batch}_fps_start"]) if int(g[f"b{batch}_fps_start"]) >= 0 else None
        counts, sums, n_kept = curriculum_update(gb, to_dev(g[f"b{batch}_active"]), Cn, R_min, R_max)
        yield batch, gb, counts, sums, n_kept, E, Cn, eval_iter


def _check_batch(g, batch, gb, counts, sums, n_kept, E, Cn, eval_iter):
    last = eval_iter - 1
    np.testing.assert_allclose(np.asarray(torch.as_tensor(gb._weight_buffer).cpu()).reshape(-1), g[f"b{batch}_weight_buffer"].reshape(-1), rtol=0, atol=1e-7)
    np.testing.assert_array_equal(np.asarray(torch.as_tensor(gb._state_buffer).cpu()), g[f"b{batch}_state_buffer"])
    assert n_kept == _scalar(g, f"b{batch}_ep{last}_add_history")
    for i in range(Cn + 1):
        assert abs(counts[i] / E - _scalar(g, f"b{batch}_ep{last}_ratio_cylinders_{i}")) < 1e-6
        assert abs((sums[i] / counts[i] if counts[i] > 0 else 0.0) - _scalar(g, f"b{batch}_ep{last}_success_cylinders_{i}")) < 1e-6
    hist = gb._history_buffer
    assert len(hist) == _scalar(g, f"b{batch}_ep{last}_history_buffer")
    np.testing.assert_array_equal(hist, g[f"b{batch}_history"])           # the same tasks in the same order: FPS picked the same indices


def test_genbuffer_and_curriculum_match_the_reference(golden):
    """`GenBuffer.insert / insert_weights / update`, the per-cylinder-count statistics, the R_min..R_max filter and `insert_history`
    incl. the normalisation (x - min) / (max - min + eps) and the farthest-point trim — against the reference's own class and its own curriculum
    statements (hideandseek_envgen.py:209-377, :1302-1333), executed by make_golden.py with an exact FPS whose start index the fixture names."""
    g = golden("g_genbuffer")
    seen = 0
    for batch, gb, counts, sums, n_kept, E, Cn, eval_iter in _replay_batches(g, lambda A, Cn, B: GenBuffer(A, Cn, seed=0, buffer_length=B)):
        _check_batch(g, batch, gb, counts, sums, n_kept, E, Cn, eval_iter)
        seen += 1
    assert seen == 2 and int(g["b1_fps_start"]) >= 0 and len(g["b1_fps_idx"]) == int(g["meta"][4])
    # the normalised cloud the reference handed to its sampler, and the selection, from this build's own pieces
    allst = np.concatenate([g["b0_history"], g["b1_state_buffer"][((g["b1_weight_buffer"] <= g["r_bounds"][1]) & (g["b1_weight_buffer"] >= g["r_bounds"][0])).reshape(-1)]])
    lo, hi = allst.min(0), allst.max(0)
    np.testing.assert_array_equal(((allst - lo) / (hi - lo + 1e-5)).astype(np.float32), g["b1_fps_normed"])
    idx = farthest_point_sampling(torch.from_numpy(g["b1_fps_normed"]), int(g["meta"][4]), start=int(g["b1_fps_start"]))
    np.testing.assert_array_equal(idx.numpy(), g["b1_fps_idx"])


def test_task_bounds_and_samplenearby_against_the_reference(golden):
    """`task_bounds` = the array the reference's `samplenearby` clips to (:320-333, its own statements executed); the reference's perturbed tasks
    (its RNG stream is unpinned, so the OUTPUTS are the fixture) pass this build's sanity check, stay inside this build's bounds and within
    expand_step (one grid cell for an expanded cylinder) of a history entry — the properties this build's own `samplenearby` is held to."""
    g = golden("g_genbuffer")
    for A, Cn in ((3, 5), (4, 5)):
        np.testing.assert_allclose(GenBuffer(A, Cn).task_bounds(), g[f"bounds_a{A}c{Cn}"], rtol=0, atol=1e-12)
    gb = GenBuffer(3, 5)
    b = gb.task_bounds()
    hist = g["near_history"].astype(np.float64)
    for expand in (0, 1):
        near = g[f"near_expand{expand}"]
        assert gb.sanity_ok(near).all()
        assert (near >= b[:, 0] - 1e-9).all() and (near <= b[:, 1] + 1e-9).all()
        lim = np.concatenate([np.full(12, 0.1), np.tile([0.2 * expand, 0.2 * expand, 0.0], 5)]) + 1e-6
        clipped = np.clip(hist, b[:, 0], b[:, 1])                          # (the history's z = 1.1..1.3 already lies in the window)
        near_some = (np.abs(near[:, None, :] - clipped[None]) <= lim).all(-1).any(-1)
        assert near_some.all()
        ours = GenBuffer(3, 5, seed=expand)
        ours.init_history(g["near_history"])
        mine = ours.samplenearby(200, expand, 0.1).astype(np.float64)
        assert ((np.abs(mine[:, None, :] - clipped[None]) <= lim).all(-1).any(-1)).all() and gb.sanity_ok(mine).all()
        if expand:                                                            # cylinders move by whole cells in both
            for arr in (near, mine):
                cyl = arr[:, 12:].reshape(-1, 5, 3)[..., :2] / 0.2
                assert np.abs(cyl - np.rint(cyl)).max() < 0.21                # = the history's own +-0.04 jitter, kept


def test_easy_cases_match_the_reference_flood(golden):
    """`init_easy_cases` with the reference's own start cells: the pursuers' cells of the reference's flood (hideandseek_envgen.py:246-262; as written it
    runs for num_agents == 4 only, which the fixture records), z inside the reference's window."""
    g = golden("g_genbuffer")
    assert g["easy_runs"].tolist() == [0, 0, 0, 1, 0]                         # A = 1..5: what the reference's method survives
    ref = g["easy_a4"]                                                        # [160, 5, 3]: pursuers, evader
    gb = GenBuffer(4, 5, seed=0)
    assert np.array_equal(gb.grid_map, g["easy_disc"])
    start = gb.to_grid(ref[:, 4, :2])
    easy = gb.init_easy_cases(start=start)
    assert easy.shape == ref.shape
    np.testing.assert_allclose(easy[..., :2], ref[..., :2], rtol=0, atol=1e-6)
    assert (np.abs(ref[..., 2] - 0.6) <= 0.1).all() and (np.abs(easy[..., 2] - 0.6) <= 0.1).all()


@pytest.mark.gpu
def test_device_genbuffer_and_episode_end_match_the_reference(golden):
    """The same fixture through the DEVICE path: `DeviceGenBuffer` (history, weights and the FPS trim `hns_fps` on the GPU) inside
    `HideAndSeek_envgen._episode_end` — the env's own success / active-cylinder / num_unif state is set to the fixture's, the statistics the
    reference's block writes (`success_buffer`, `success_unif`, `ratio_cylinders_i`, `success_cylinders_i`, `add_history`, `history_buffer`,
    `ratio_unif`) and the history are compared after every episode."""
    from hns_amd import abi
    from hns_amd.envgen import DeviceGenBuffer, HideAndSeek_envgen
    g = golden("g_genbuffer")
    A, Cn, E, eval_iter, B = (int(x) for x in g["meta"])
    R_min, R_max = (float(x) for x in g["r_bounds"])
    cfg = config.make_cfg({"name": "HideAndSeek_envgen", "num_agents": A, "ratio_unif": 0.3, "eval_iter": eval_iter, "R_min": R_min, "R_max": R_max,
                           "use_particle_generator": 1, "cylinder": {"max_num": Cn, "min_num": 2}, "env": {"num_envs": E, "max_episode_length": 5}})
    env = HideAndSeek_envgen(cfg)
    env.set_seed(0)
    env.reset()
    env.gen_buffer = DeviceGenBuffer(env, buffer_length=B, seed=0)
    dev = env.device
    srow = abi.STAT_NAMES.index("success")
    for batch in range(2):
        env._tasks_dev.copy_(torch.from_numpy(g[f"b{batch}_tasks"]))
        env.num_unif = int(g[f"b{batch}_num_unif"])
        env.active_cylinders = torch.from_numpy(g[f"b{batch}_active"]).to(dev)
        env.gen_buffer.insert(env._tasks_dev)
        env.gen_buffer.fps_start = int(g[f"b{batch}_fps_start"]) if int(g[f"b{batch}_fps_start"]) >= 0 else None
        env.update_iter = 0
        for ep in range(eval_iter):
            env._bufs["stats"][srow].copy_(torch.from_numpy(g[f"b{batch}_success{ep}"]).reshape(-1))
            env._episode_end()
            assert env.update_iter == int(g[f"b{batch}_ep{ep}_update_iter"])
            for k in ["success_buffer", "success_unif", "history_buffer", "add_history", "ratio_unif"] + [f"{n}_cylinders_{i}" for n in ("ratio", "success") for i in range(Cn + 1)]:
                want = np.asarray(g[f"b{batch}_ep{ep}_{k}"], dtype=np.float64)
                got = env.stats[k].cpu().numpy().astype(np.float64)
                np.testing.assert_allclose(got, np.broadcast_to(want, got.shape), rtol=0, atol=1e-6, err_msg=f"batch {batch} episode {ep}: stats.{k}")
        np.testing.assert_array_equal(env.gen_buffer._history_buffer, g[f"b{batch}_history"])
        np.testing.assert_allclose(env.gen_buffer._weight_buffer.cpu().numpy(), g[f"b{batch}_weight_buffer"].reshape(-1), rtol=0, atol=1e-7)
    # success above the threshold: uniform tasks only from then on (:1303-1304)
    env.success_threshold = 0.5
    env._bufs["stats"][srow].fill_(1.0)
    env.gen_buffer.insert(env._tasks_dev)
    env._episode_end()
    assert env.ratio_unif == float(g["ratio_unif_after_threshold"]) == 1.0


# ---- what the generator env leaves behind in corner cases ------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_construction_leaves_the_generator_empty_and_partial_resets_archive_live_placements():
    """(1) `_prewarm` runs the real update path on stand-in batches: afterwards the generator holds what it held before (no fabricated task batch or weights).
    (2) A PARTIAL reset at a batch boundary archives, for the envs that keep running, the placement they are in — not the stale content of their task rows.
    (3) The mask of a collector — a BOOL view of the env's done buffer — is recognised as that buffer (the lock-step fast path of `_note_reset`)."""
    import torch
    from hns_amd.envgen import HideAndSeek_envgen
    from hns_amd.tensordict_shim import TensorDict
    E, L = 256, 6
    cfg = config.make_cfg({"name": "HideAndSeek_envgen", "num_agents": 3, "eval_iter": 2, "R_min": 0.0, "R_max": 1.0, "ratio_unif": 0.3,
                           "use_particle_generator": 1, "cylinder": {"max_num": 8, "min_num": 8}, "env": {"num_envs": E, "max_episode_length": L}})
    env = HideAndSeek_envgen(cfg)
    g = env.gen_buffer
    assert g._state_buffer.shape[0] == 0 and g._weight_buffer.numel() == 0 and g._temp_state is None and g._temp_weights == [] and len(g) == 0
    env.set_seed(2)
    env.reset()
    assert env.update_iter == 0
    # a partial reset while update_iter == 0: half of the envs are reset, the others keep running where they are
    for _ in range(2):
        env.step(env.rand_step_input())
    b = env._bufs
    live = torch.cat([b["drone_state"][..., 0:3].reshape(E, -1), b["target_pos"].reshape(E, -1), b["cylinders"].reshape(E, -1)], dim=1).clone()
    env._tasks_dev.fill_(123.0)                                       # whatever the rows held
    mask = torch.arange(E, device=env.device) % 2 == 0
    env.reset(TensorDict({"_reset": mask}, env.batch_size))
    archived = g._temp_state                                          # what `insert` took
    assert archived is not None and archived.shape == (E, g.task_dim)
    keep = ~mask
    assert torch.equal(archived[keep][:, :], live[keep]), "an env that was not reset is archived with something else than its placement"
    assert not (archived[mask] == 123.0).any()                        # reset envs: the placement as sampled (written by the kernel)
    # the collector's mask: a bool view of the done buffer
    env2 = HideAndSeek_envgen(cfg)
    env2.set_seed(3)
    env2.reset()
    for _ in range(L):
        env2.step(env2.rand_step_input())
    assert env2._all_done and env2._since_full_reset >= L
    done_view = env2._bufs["done"].view(torch.bool)
    env2.reset(TensorDict({"_reset": done_view}, env2.batch_size))
    assert env2._reset_with_done_buffer and env2._since_full_reset == 0 and not env2._all_done
