"""HideAndSeek_envgen — the Adaptive Environment Generator variant (BASELINE config 4).

The per-step math is the HideAndSeek step kernel unchanged; what differs is per-episode
bookkeeping on the host (reference omni_drones/envs/hide_and_seek/hideandseek_envgen.py):
  * `GenBuffer` (:209-377): a history of task vectors `[drone xyz * A | evader xyz | cylinder xyz * C]`,
    `samplenearby` perturbation with a grid sanity check, success-weighted insertion and
    farthest-point-sampling (FPS) trimming to 5000 entries;
  * `_reset_idx` (:875-902): every `eval_iter` episodes a new task batch = `ratio_unif` uniform tasks
    + perturbed buffer tasks, replayed for `eval_iter` episodes;
  * the curriculum block at episode end (:1302-1333) and the extra statistics (:617-651, :1241-1246).

In the env the buffer lives on the device (`DeviceGenBuffer`): the history, the task batch and the
success weights never leave HBM; `samplenearby` is `hns_perturb_tasks` (one thread per task) and the
FPS trim is `hns_fps` (one persistent launch, 5 ms for 5000 of 70 000 tasks since round 5) — csrc/hns_envgen.hip.
`GenBuffer` below is the same logic on the host in numpy (the reference's class surface; used by the
CPU tests and as the plain restatement of the device path).

Differences from the reference, all distribution-preserving (the generator's RNG streams
and DGL's FPS start point are unpinned):
  * `samplenearby` perturbs every task independently (Philox stream per task, a fresh history entry
    per attempt, the entry itself after 10 failed attempts) instead of a Python loop per task that
    pads failures with copies of other tasks' results — needed at 65 536 envs;
  * FPS is the build's own (`hns_fps`; start index random, ties -> lower index) instead of
    `dgl.geometry.farthest_point_sampler` (not installed; its version is unpinned in the reference);
  * task placement on the device goes through `hns_reset_tasks`; uniform tasks are sampled by the
    reset kernel itself and copied into the task batch on the device;
  * `success_buffer` / `success_unif` are refreshed at episode end (where `EpisodeStats` reads them)
    rather than every step, which would need a cross-env reduction per step.
"""
import ctypes as C
import math

import numpy as np
import torch

from .env import HideAndSeek, HnsError


def farthest_point_sampling(points, k, start=0):
    """Indices of k points chosen by iterative farthest-point sampling (torch, any device)."""
    n = points.shape[0]
    if k >= n:
        return torch.arange(n, device=points.device)
    idx = torch.empty(k, dtype=torch.long, device=points.device)
    dist = torch.full((n,), float("inf"), device=points.device, dtype=points.dtype)
    cur = torch.tensor(int(start) % n, device=points.device)
    for i in range(k):
        idx[i] = cur
        d = ((points - points[cur]) ** 2).sum(-1)
        dist = torch.minimum(dist, d)
        dist[cur] = -1.0                            # chosen points leave the pool (distinct indices among duplicates)
        cur = torch.argmax(dist)
    return idx


class GenBuffer:
    """hideandseek_envgen.py:209-377."""

    def __init__(self, num_agents, num_cylinders, device="cpu", arena_size=0.9, cylinder_size=0.1, max_height=1.2,
                 buffer_length=5000, seed=0, num_targets=1):
        self.num_agents, self.num_cylinders, self.device = num_agents, num_cylinders, device
        self.num_targets = int(num_targets)                           # 2: the two-evader extension — [pursuers | evader 0 | evader 1 | cylinders]
        self.task_dim = 3 * num_agents + 3 * self.num_targets + 3 * num_cylinders       # reference: 18 + 3A (C = 5)
        self._state_buffer = np.zeros((0, 1), dtype=np.float32)
        self._history_buffer = np.zeros((0, self.task_dim), dtype=np.float32)
        self._weight_buffer = np.zeros((0, 1), dtype=np.float32)
        self.buffer_length = buffer_length
        self.eps = 1e-5
        self.update_method = "fps"
        self._temp_state_buffer = []
        self._temp_weight_buffer = []
        self.arena_size, self.cylinder_size, self.max_height = arena_size, cylinder_size, max_height
        self.grid_size = 2 * cylinder_size
        self.num_grid = int(arena_size * 2 / self.grid_size)
        self.boundary = arena_size - 0.1
        half = self.num_grid // 2
        ii, jj = np.meshgrid(np.arange(self.num_grid), np.arange(self.num_grid), indexing="ij")
        self.grid_map = (np.sqrt((ii - half) ** 2 + (jj - half) ** 2) >= half).astype(np.int64)   # :168-181
        self.rng = np.random.default_rng(seed)
        self.fps_start = None             # start index of the next farthest-point trim; None = drawn at random (DGL draws it at random too)

    # -- grid helpers (:121-164) ------------------------------------------------------------------
    def to_grid(self, xy):
        g = np.rint(np.asarray(xy, dtype=np.float64) / self.grid_size).astype(np.int64) + self.num_grid // 2
        return np.clip(g, 0, self.num_grid - 1)

    def sanity_ok(self, tasks):
        """:187-207 vectorised: every object sits in its own free cell of the disc."""
        A, Cn = self.num_agents, self.num_cylinders
        n = tasks.shape[0]
        xyz = tasks.reshape(n, A + self.num_targets + Cn, 3)
        g = self.to_grid(xyz[..., :2])
        cell = g[..., 0] * self.num_grid + g[..., 1]
        free = self.grid_map.reshape(-1)[cell] == 0
        cs = np.sort(cell, axis=1)
        distinct = (np.diff(cs, axis=1) != 0).all(axis=1)
        return free.all(axis=1) & distinct

    # -- buffer maintenance ---------------------------------------------------------------------------
    def init_history(self, init_tasks):
        self._history_buffer = np.asarray(init_tasks, dtype=np.float32).reshape(-1, self.task_dim)

    def _flood_rank(self):
        """Discovery rank of every offset (dx, dy) in a 4-neighbour breadth-first flood from the origin, neighbours taken in the
        order -x, +x, -y, +y (hideandseek_envgen.py:246-262).  The reference floods THROUGH occupied cells, so nothing blocks and
        the flood on the bounded grid discovers the in-grid cells in the same relative order as on the open plane (every
        shortest-path predecessor of an in-grid cell lies between it and the start).  One table serves every start cell.
        Built ring by ring: a ring's cells are discovered in the order their first parent sits in the previous ring."""
        n = self.num_grid
        rank = np.full((2 * n + 1, 2 * n + 1), -1, dtype=np.int64)
        steps = np.array([(-1, 0), (1, 0), (0, -1), (0, 1)])
        ring, count = np.array([[0, 0]]), 1
        rank[n, n] = 0
        while len(ring):
            cand = (ring[:, None, :] + steps[None, :, :]).reshape(-1, 2)          # parents in ring order, their neighbours in step order
            cand = cand[(np.abs(cand) <= n).all(axis=1)]
            fresh = cand[rank[cand[:, 0] + n, cand[:, 1] + n] < 0]
            _, first = np.unique(fresh[:, 0] * (4 * n + 4) + fresh[:, 1], return_index=True)
            ring = fresh[np.sort(first)]                                            # first appearance wins, order kept
            rank[ring[:, 0] + n, ring[:, 1] + n] = count + np.arange(len(ring))
            count += len(ring)
        return rank

    def init_easy_cases(self, start=None):
        """Easy starting tasks (hideandseek_envgen.py:235-277): the evader on a random free cell (`start`: given cells [B, 2] instead of
        drawn ones), the pursuers on the free cells the flood from that cell reaches first.  All samples at once: rank every free cell by the
        flood table, keep the num_agents smallest per sample.  (As written the reference's method runs for num_agents == 4 only — with fewer
        pursuers its `found` list holds up to four cells and `np.array` of the ragged rows / the concat with the z column fails,
        tests/golden/make_golden.py::gen_genbuffer; taking the first num_agents cells in discovery order is this build's reading of the intent.)"""
        if self.num_targets != 1:
            raise NotImplementedError("use_init_easy places one evader (the reference's flood, hideandseek_envgen.py:235-277)")
        n, A, B = self.num_grid, self.num_agents, self.buffer_length
        free = np.argwhere(self.grid_map == 0)                                     # [F, 2]
        if start is None:
            start = free[np.array([self.rng.integers(len(free)) for _ in range(B)])]   # one draw per sample, in sample order
        else:
            start = np.asarray(start, dtype=np.int64).reshape(-1, 2)
            B = start.shape[0]
        if A > 4:
            return self._easy_cases_literal(start)
        off = free[None, :, :] - start[:, None, :]                                 # [B, F, 2]
        rank = self._flood_rank()[off[..., 0] + n, off[..., 1] + n]                # [B, F]; the start itself has rank 0
        rank[rank == 0] = np.iinfo(np.int64).max                                   # the evader's cell is not a pursuer's
        nearest = np.argsort(rank, axis=1, kind="stable")[:, :A]                   # [B, A] indices into `free`, in discovery order
        cells = np.concatenate([free[nearest], start[:, None, :]], axis=1).astype(np.float64)   # pursuers, then the evader
        xy = np.clip((cells - n // 2) * self.grid_size, -self.boundary, self.boundary)
        z = (self.rng.random((B, A + 1, 1)) * 0.2 - 0.1) + self.max_height / 2
        return np.concatenate([xy, z], axis=-1).astype(np.float32)

    def _easy_cases_literal(self, start):
        """More than four pursuers: the reference's queue flood as written (hideandseek_envgen.py:246-262), because of what its
        `if len(found) == 4: break` does there — the fourth free cell found is never enqueued and the remaining neighbours of the
        cell being expanded are skipped, which changes which cells the fifth and later pursuers get near the arena rim.  (Up to
        four pursuers the flood ends at that `break` and the rank table above gives the same cells.)  Where the reference's list ends up
        with MORE than num_agents cells (it appends every free neighbour it visits) its `np.array(...)` of ragged rows fails; taking the
        first num_agents in discovery order is this build's choice for that case, as is the error below when the flood ends with fewer."""
        from collections import deque
        n, A, B = self.num_grid, self.num_agents, start.shape[0]
        cells = np.zeros((B, A + 1, 2), dtype=np.float64)
        for k in range(B):
            x, y = int(start[k, 0]), int(start[k, 1])
            visited = np.zeros((n, n), dtype=bool)
            queue = deque([(x, y)])
            visited[x, y] = True
            found = []
            while queue and len(found) < A:
                cx, cy = queue.popleft()
                for dx, dy in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                    nx, ny = cx + dx, cy + dy
                    if 0 <= nx < n and 0 <= ny < n and not visited[nx, ny]:
                        visited[nx, ny] = True
                        if self.grid_map[nx, ny] == 0:
                            found.append((nx, ny))
                            if len(found) == 4:
                                break
                        queue.append((nx, ny))
            if len(found) < A:
                raise ValueError(f"init_easy_cases: the flood from cell ({x}, {y}) reached only {len(found)} free cells for {A} pursuers "
                                 "(the reference's `if len(found) == 4: break` skips the rest of the cell being expanded; use fewer pursuers or a larger grid)")
            cells[k, :A] = found[:A]
            cells[k, A] = (x, y)
        xy = np.clip((cells - n // 2) * self.grid_size, -self.boundary, self.boundary)
        z = (self.rng.random((B, A + 1, 1)) * 0.2 - 0.1) + self.max_height / 2
        return np.concatenate([xy, z], axis=-1).astype(np.float32)

    def insert(self, states):
        self._temp_state_buffer.extend(np.array(states, dtype=np.float32, copy=True))

    def insert_weights(self, weights):
        self._temp_weight_buffer.append(np.asarray(weights, dtype=np.float32).reshape(-1, 1))

    def update(self):
        self._state_buffer = np.array(self._temp_state_buffer)
        self._weight_buffer = np.stack(self._temp_weight_buffer, axis=-1).mean(-1)
        self._temp_state_buffer, self._temp_weight_buffer = [], []

    def insert_history(self, states):
        states = np.asarray(states, dtype=np.float32).reshape(-1, self.task_dim)
        if len(states) == 0:
            return
        all_states = np.concatenate([self._history_buffer, states])
        if self.update_method == "fifo":
            self._history_buffer = all_states[-self.buffer_length:]
        elif all_states.shape[0] > self.buffer_length:
            lo, hi = all_states.min(0), all_states.max(0)
            normed = torch.as_tensor((all_states - lo) / (hi - lo + self.eps), device=self.device)
            start = int(self.rng.integers(all_states.shape[0])) if self.fps_start is None else int(self.fps_start)
            idx = farthest_point_sampling(normed, self.buffer_length, start=start)
            self._history_buffer = all_states[idx.cpu().numpy()]
        else:
            self._history_buffer = all_states

    def task_bounds(self):
        """:320-333 (incl. the reference's z window [max_height-0.1, max_height+0.1] for drones/evader)."""
        cb = int(self.arena_size / self.grid_size) * self.grid_size
        bxy = self.arena_size / math.sqrt(2.0) - 0.1
        drone = [[-bxy, bxy], [-bxy, bxy], [self.max_height - 0.1, self.max_height + 0.1]]
        cyl = [[-cb, cb], [-cb, cb], [-20.0, self.max_height / 2]]
        return np.array(drone * (self.num_agents + self.num_targets) + cyl * self.num_cylinders)

    def samplenearby(self, num_tasks, expand_cylinders, expand_step):
        """:316-370, vectorised: perturb, clip, sanity-check, retry the failures (<= 10 rounds)."""
        idx = self.rng.integers(self._history_buffer.shape[0], size=num_tasks)
        origin = self._history_buffer[idx].astype(np.float64)
        b = self.task_bounds()
        nd = self.task_dim - 3 * self.num_cylinders
        out = np.zeros_like(origin)
        done = np.zeros(num_tasks, dtype=bool)
        for _ in range(10):
            todo = np.flatnonzero(~done)
            if todo.size == 0:
                break
            noise = np.zeros((todo.size, self.task_dim))
            noise[:, :nd] = self.rng.uniform(-1, 1, size=(todo.size, nd)) * expand_step
            if expand_cylinders:
                cn = np.zeros((todo.size, self.num_cylinders, 3))
                cn[..., :2] = self.rng.choice([-1, 0, 1], size=(todo.size, self.num_cylinders, 2)) * self.grid_size
                noise[:, nd:] = cn.reshape(todo.size, -1)
            cand = np.clip(origin[todo] + noise, b[:, 0], b[:, 1])
            ok = self.sanity_ok(cand)
            out[todo[ok]] = cand[ok]
            done[todo[ok]] = True
        good = out[done]
        if good.shape[0] == 0:
            raise ValueError("samplenearby: no perturbed task passed the grid sanity check")
        if good.shape[0] < num_tasks:                                      # :363-368
            add = good[self.rng.integers(good.shape[0], size=num_tasks - good.shape[0])]
            good = np.concatenate([add, good])
        return good.astype(np.float32)

    def sample(self, num_tasks):
        return self._history_buffer[self.rng.integers(self._history_buffer.shape[0], size=num_tasks)]

    def save_task(self, model_dir, episode):
        np.save("{}/history_{}.npy".format(model_dir, episode), self._history_buffer)


class DeviceGenBuffer:
    """GenBuffer (hideandseek_envgen.py:209-377) with every array resident on the env's GPU."""

    def __init__(self, env, buffer_length=5000, seed=0):
        self.env, self.lib, self.device = env, env._lib, env.device
        self.num_agents, self.num_cylinders, self.num_targets = env.num_agents, env.num_cylinders, env.num_targets
        self.task_dim = 3 * self.num_agents + 3 * self.num_targets + 3 * self.num_cylinders
        self.buffer_length, self.eps, self.update_method = buffer_length, 1e-5, "fps"
        self._history = torch.zeros(0, self.task_dim, device=self.device)
        self._state_buffer = torch.zeros(0, self.task_dim, device=self.device)
        self._weight_buffer = torch.zeros(0, device=self.device)
        self._temp_state, self._temp_weights = None, []
        self._fps_scratch = torch.zeros(int(self.lib.hns_fps_scratch_bytes()), dtype=torch.uint8, device=self.device)
        self._fps_idx = torch.zeros(buffer_length, dtype=torch.int32, device=self.device)
        self.rng = np.random.default_rng(seed)
        self.fps_start = None             # as GenBuffer.fps_start
        self._draws = 0

    # numpy views for callers that want the reference's attributes (statistics, save_task, tests)
    @property
    def _history_buffer(self):
        return self._history.cpu().numpy()

    def __len__(self):
        return int(self._history.shape[0])

    def init_history(self, init_tasks):
        self._history = torch.as_tensor(np.asarray(init_tasks, dtype=np.float32).reshape(-1, self.task_dim), device=self.device)

    def insert(self, states):
        self._temp_state = states.detach().clone()

    def insert_weights(self, weights):
        self._temp_weights.append(weights.detach().reshape(-1).clone())

    def update(self):
        self._state_buffer = self._temp_state
        self._weight_buffer = torch.stack(self._temp_weights, dim=-1).mean(-1)
        self._temp_state, self._temp_weights = None, []

    def insert_history(self, states):
        if states.shape[0] == 0:
            return
        all_states = torch.cat([self._history, states.reshape(-1, self.task_dim)]).contiguous()
        n = int(all_states.shape[0])
        if self.update_method == "fifo":
            self._history = all_states[-self.buffer_length:].contiguous()
        elif n > self.buffer_length:
            lo, hi = all_states.min(0).values, all_states.max(0).values
            normed = ((all_states - lo) / (hi - lo + self.eps)).contiguous()
            start = int(self.rng.integers(n)) if self.fps_start is None else int(self.fps_start)
            rc = self.lib.hns_fps(normed.data_ptr(), n, self.task_dim, self.buffer_length, start,
                                  self._fps_idx.data_ptr(), self._fps_scratch.data_ptr(), self.env._stream())
            self.env._check(rc, "hns_fps")
            self._history = all_states.index_select(0, self._fps_idx.long())
            if int(self._fps_scratch[:8].view(torch.int64)[0]) != 0:          # also the sync that ends the launch
                raise HnsError("hns_fps gave up: a workgroup of the persistent launch never became resident")
        else:
            self._history = all_states

    def samplenearby_into(self, out, expand_cylinders, expand_step):
        """Fill `out` ([n, task_dim], a device tensor or a contiguous slice of one) with perturbed history tasks."""
        assert out.is_contiguous() and out.shape[1] == self.task_dim
        self._draws += 1
        seed = (int(self.env.seed) * 0x9E3779B97F4A7C15 + self._draws) & 0xFFFFFFFFFFFFFFFF
        rc = self.lib.hns_perturb_tasks(self.env._env, self._history.data_ptr(), len(self), out.data_ptr(), int(out.shape[0]),
                                        int(bool(expand_cylinders)), C.c_float(expand_step), C.c_uint64(seed), self.env._stream())
        self.env._check(rc, "hns_perturb_tasks")

    def samplenearby(self, num_tasks, expand_cylinders, expand_step):
        out = torch.zeros(num_tasks, self.task_dim, device=self.device)
        self.samplenearby_into(out, expand_cylinders, expand_step)
        return out.cpu().numpy()

    def sample(self, num_tasks):
        idx = torch.as_tensor(self.rng.integers(len(self), size=num_tasks), device=self.device)
        return self._history.index_select(0, idx).cpu().numpy()

    def save_task(self, model_dir, episode):
        np.save("{}/history_{}.npy".format(model_dir, episode), self._history_buffer)


def curriculum_update(gen_buffer, active_cylinders, num_cylinders, R_min, R_max):
    """The generator's update at the end of every `eval_iter`-th episode (hideandseek_envgen.py:1311-1333): `gen_buffer.update()` (the mean success
    weight of every task over the episodes it was replayed), the per-cylinder-count statistics, the tasks whose weight lies in [R_min, R_max]
    inserted into the history (trimmed by farthest-point sampling).  `gen_buffer` is a DeviceGenBuffer (tensors on the env's GPU) or the host
    GenBuffer (numpy); returns (count of tasks per number of active cylinders [C+1], sum of their weights [C+1] — fp64 tensors on the buffer's device,
    nothing is read back for them — and the number of tasks kept)."""
    gen_buffer.update()
    w = torch.as_tensor(gen_buffer._weight_buffer).reshape(-1)
    act = torch.as_tensor(active_cylinders).reshape(-1).long().to(w.device)
    # (a one-hot sum instead of torch.bincount — which reads the largest index back to size its output, a host synchronisation per call — and instead of
    #  index_add_, whose 65 536 fp64 atomics on nine addresses took 5 ms)
    onehot = (act.unsqueeze(1) == torch.arange(num_cylinders + 1, device=w.device).unsqueeze(0)).double()        # [n, C + 1]
    counts = onehot.sum(0)
    sums = (onehot * w.double().unsqueeze(1)).sum(0)
    keep = (w <= R_max) & (w >= R_min)
    states = gen_buffer._state_buffer
    kept = states[keep] if isinstance(states, torch.Tensor) else states[keep.cpu().numpy()]      # (the one read-back: how many tasks are kept)
    gen_buffer.insert_history(kept)                                   # (global mode: every rank takes part, also with nothing to add)
    return counts, sums, int(kept.shape[0])


class HideAndSeek_envgen(HideAndSeek):
    def __init__(self, cfg, headless=True, env_index_offset=0, write_critic_state=None):
        super().__init__(cfg, headless, env_index_offset, write_critic_state)
        t = cfg.task
        self.use_particle_generator = int(t.get("use_particle_generator", 1))
        self.ratio_unif = float(t.get("ratio_unif", 0.3))
        self.eval_iter = int(t.get("eval_iter", 3))
        self.R_min, self.R_max = float(t.get("R_min", 0.5)), float(t.get("R_max", 0.9))
        self.success_threshold = float(t.get("success_threshold", 1.0))
        self.expand_cylinders, self.expand_step = int(t.get("expand_cylinders", 0)), float(t.get("expand_step", 0.1))
        self.use_init_easy = int(t.get("use_init_easy", 0))
        self.update_iter = 0
        self.num_unif = self.num_envs
        A, Cn, E = self.num_agents, self.num_cylinders, self.num_envs
        self.gen_buffer = DeviceGenBuffer(self, seed=int(cfg.get("seed", 0)))
        # task.global_gen_buffer: one history for the whole data-parallel job, owned by rank 0 (the reference's semantics under
        # sharding, sharding.GlobalGenBuffer); default: every rank keeps its own
        self.global_gen_buffer = False
        if int(t.get("global_gen_buffer", 0)):
            import torch.distributed as dist
            from . import sharding
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                self.gen_buffer = sharding.GlobalGenBuffer(self.gen_buffer, int(t.get("num_envs_total", 0)) or None)
                self.global_gen_buffer = True
        self._env_offset = int(env_index_offset)
        if self.use_init_easy:                                                # :485-495
            easy = GenBuffer(A, Cn, arena_size=float(t.arena_size), cylinder_size=float(t.cylinder.size),
                             max_height=float(t.max_height), seed=int(cfg.get("seed", 0)), num_targets=self.num_targets).init_easy_cases()
            cyl = np.tile(np.array([0.0, 0.0, -20.0], np.float32), (easy.shape[0], Cn, 1))
            self.gen_buffer.init_history(np.concatenate([easy.reshape(easy.shape[0], -1), cyl.reshape(easy.shape[0], -1)], axis=1))
        self.task_dim = self.gen_buffer.task_dim
        self._tasks_dev = torch.zeros(E, self.task_dim, device=self.device)
        self.active_cylinders = torch.zeros(E, 1, device=self.device)
        extra = ["success_buffer", "success_unif", "history_buffer", "add_history", "ratio_unif"]
        extra += [f"ratio_cylinders_{i}" for i in range(Cn + 1)] + [f"success_cylinders_{i}" for i in range(Cn + 1)]
        # :617-651 — rows of ONE [n, E] buffer (a statistic is filled by one row assignment, the whole set cloned by one copy)
        self._extra_buf = torch.zeros(len(extra), E, device=self.device)
        self._extra_row = {k: i for i, k in enumerate(extra)}
        self._extra = {k: self._extra_buf[i].unsqueeze(-1) for i, k in enumerate(extra)}
        for k, v in self._extra.items():
            self.stats.set(k, v)
        self.generator_seconds = 0.0
        self._prewarm()

    def _prewarm(self):
        """Run the two generator kernels once on a toy problem: the first launch of a kernel pays its code-object load
        (the first task batch used to cost ~240 ms, later ones 26 ms); construction is the place for that."""
        g = self.gen_buffer
        toy = torch.rand(96, g.task_dim, device=self.device).contiguous()
        idx = torch.zeros(32, dtype=torch.int32, device=self.device)
        self._check(self._lib.hns_fps(toy.data_ptr(), 96, g.task_dim, 32, 0, idx.data_ptr(), g._fps_scratch.data_ptr(), self._stream()), "hns_fps")
        out = torch.zeros(8, g.task_dim, device=self.device)
        self._check(self._lib.hns_perturb_tasks(self._env, toy.data_ptr(), 96, out.data_ptr(), 8, 0, C.c_float(0.1), C.c_uint64(1), self._stream()),
                    "hns_perturb_tasks")
        torch.cat([toy, toy]).index_select(0, idx.long()).min(0)          # the torch ops of the trim (allocator, kernels)
        # and one trim at the size of a real update (history + every env's task), so the allocator holds those blocks
        # (everything the stand-in batches below touch is put back at the end: after construction the generator holds what it held before — ADVICE r5)
        keep = g._history
        kept = {a: getattr(g, a) for a in ("_state_buffer", "_weight_buffer", "_temp_state", "_temp_weights") if hasattr(g, a)}
        kept["_temp_weights"] = list(kept.get("_temp_weights") or [])
        g._history = torch.rand(g.buffer_length, g.task_dim, device=self.device)
        g.insert_history(torch.rand(self.num_envs, g.task_dim, device=self.device))
        # ... and the shape of the FIRST real update (empty history + every env's task: another launch shape of the trim), with
        # the statistics of _episode_end, so that no first-use cost is left for the first task batch (it read 39 ms against 15 ms
        # for the later ones in round 3's bench, 112 ms on the driver's box of round 2)
        # (the real code path — curriculum_update on a stand-in batch, the row assignments of _episode_end — so that every torch kernel it launches has
        #  been launched once: the first real update read 26 ms against 6.7 ms when only the trim had been warmed)
        g._history = torch.zeros(0, g.task_dim, device=self.device)
        g.insert(torch.rand(self.num_envs, g.task_dim, device=self.device))
        w = torch.rand(self.num_envs, 1, device=self.device)
        g.insert_weights(w)
        counts, sums, _ = curriculum_update(g, torch.zeros(self.num_envs, 1, device=self.device), self.num_cylinders, 0.0, 1.0)
        buf, row, Cn = self._extra_buf, self._extra_row, self.num_cylinders
        flat = w.reshape(-1)
        rates = torch.stack([flat[1:].mean(), flat[:1].mean(), flat.mean()])
        buf[row["success_buffer"]] = rates[0]
        buf[row["success_unif"]] = flat
        float(rates[2])
        r0 = row["ratio_cylinders_0"]
        buf[r0:r0 + Cn + 1] = (counts / self.num_envs).float().unsqueeze(1)
        buf[r0:r0 + Cn + 1] = torch.where(counts > 0, sums / counts.clamp(min=1.0), torch.zeros_like(sums)).float().unsqueeze(1)
        buf[row["add_history"]] = 1.0
        buf.zero_()
        int(self._bufs["done"].sum(dtype=torch.int32))
        self._clone_stats()
        g._history = keep
        for a, v in kept.items():
            setattr(g, a, v)
        torch.cuda.synchronize(self.device)

    @property
    def all_tasks(self):
        """The current task batch [E, task_dim] as numpy (hideandseek_envgen.py:886-899 keeps it on the host)."""
        return self._tasks_dev.cpu().numpy()

    # ---- hideandseek_envgen.py:875-902 ------------------------------------------------------------------
    def _reset(self, tensordict=None, **kwargs):
        if not (self.use_particle_generator and int(self.cfg.task.use_random_cylinder)):
            return super()._reset(tensordict, **kwargs)
        import time
        mask_t = None
        self._reset_with_done_buffer = False
        if tensordict is not None and "_reset" in tensordict.keys():
            orig = tensordict.get("_reset")
            # a collector builds `_reset` from `next.done` — a BOOL view of the env's done buffer, which the conversion below copies: whether the mask IS that buffer
            # is decided on the original tensor's storage (compared after the conversion the fast path of `_note_reset` only ever fired for bench.py's uint8 view)
            self._reset_with_done_buffer = (orig.numel() == self.num_envs and orig.untyped_storage().data_ptr() == self._bufs["done"].untyped_storage().data_ptr()
                                            and orig.storage_offset() * orig.element_size() == self._bufs["done"].storage_offset())
            mask_t = orig.reshape(self.num_envs).to(torch.uint8).contiguous()
        last_stats = self._clone_stats()
        E = self.num_envs
        t0 = time.perf_counter()
        mptr = C.c_void_p(mask_t.data_ptr()) if mask_t is not None else None
        if self.update_iter == 0:
            hist = len(self.gen_buffer)
            num_buffer = min(hist, int(E * (1 - self.ratio_unif)))
            if self.global_gen_buffer:
                num_buffer = self.gen_buffer.buffer_share(E, self._env_offset, self.ratio_unif)
            self.num_unif = E - num_buffer
            if num_buffer > 0:
                self.gen_buffer.samplenearby_into(self._tasks_dev[self.num_unif:], self.expand_cylinders, self.expand_step)
            partial = mask_t is not None and not (getattr(self, "_all_done", False) and self._reset_with_done_buffer)
            if partial:
                # a PARTIAL reset at a batch boundary (the reference resets every env there, :875-902): the kernel writes rows of MASKED envs only, so an env
                # that keeps running would be archived with whatever its row held (zeros, an older task).  Those rows get the placement the env is in — taken
                # BEFORE the reset's extra physics step moves the scene, as the sampled rows are
                b = self._bufs
                live = torch.cat([b["drone_state"][..., 0:3].reshape(E, -1), b["target_pos"].reshape(E, -1), b["cylinders"].reshape(E, -1)], dim=1)
            self._check(self._lib.hns_reset_tasks(self._env, mptr, C.c_void_p(self._tasks_dev.data_ptr()),
                                                  C.c_int32(self.num_unif), C.c_uint64(self.seed), self._stream()), "hns_reset_tasks")
            # the uniform tasks were sampled by the reset kernel, which wrote them into the rows below num_unif as SAMPLED — before the extra
            # physics step of task.reset_extra_step moves the bodies (the reference archives `tasks_unif`, :883-895, and steps afterwards, :1013)
            if partial:
                stale = (mask_t == 0) & (torch.arange(E, device=self.device) < self.num_unif)
                self._tasks_dev.copy_(torch.where(stale.unsqueeze(1), live, self._tasks_dev))
            self.gen_buffer.insert(self._tasks_dev)
        else:
            self._check(self._lib.hns_reset_tasks(self._env, mptr, C.c_void_p(self._tasks_dev.data_ptr()),
                                                  C.c_int32(0), C.c_uint64(self.seed), self._stream()), "hns_reset_tasks")
        self.active_cylinders = (self._bufs["cylinders"][..., 2] > 0.0).float().sum(-1, keepdim=True)   # :902
        self.generator_seconds += time.perf_counter() - t0
        self._keep_mask = mask_t
        self._state_version += 1
        self._note_reset(mask_t)
        self._needs_reset = False
        if self.use_TP_net:
            self._tp_observe()
        td = self._obs_tensordict()
        if not self.training:
            td = self._fresh_obs(td)              # eval mode: new tensors, as in the base class's `_reset`
        td.set("stats", last_stats)
        td.set("truncated", (self.progress_buf > self.max_episode_length).unsqueeze(1))
        return td

    def _episode_mirror_needed(self):
        return bool(self.use_particle_generator) or super()._episode_mirror_needed()

    def _note_reset(self, mask_t):
        """Episodes run in lock step: when the last step saw EVERY env done and the reset's mask is the env's own `done` buffer (what the step wrote, what
        the reset kernel reads), every env starts over — known without reading max(progress) back."""
        all_done, self._all_done = getattr(self, "_all_done", False), False          # (consumed: a later reset must not inherit it)
        if mask_t is not None and all_done and getattr(self, "_reset_with_done_buffer", False):
            self._since_full_reset = 0
            return
        super()._note_reset(mask_t)

    def import_state(self, arrays, check=False):
        self._all_done = False                     # a restored state says nothing about the step before it
        return super().import_state(arrays, check)

    # ---- curriculum at episode end, hideandseek_envgen.py:1241-1246, 1302-1333 ----------------------------
    def _step(self, tensordict):
        out = super()._step(tensordict)
        if self.use_particle_generator and self._since_full_reset >= self.max_episode_length:
            n_done = int(self._bufs["done"].sum(dtype=torch.int32))        # ONE read-back: is any env done, and are they all
            self._all_done = n_done == self.num_envs
            if n_done:
                self._episode_end()
        else:
            self._all_done = False
        return out

    def _episode_end(self):
        """hideandseek_envgen.py:1241-1246, :1302-1336 at the end of an episode.  Host synchronisations: ONE read-back of the three success rates
        (the threshold decision below needs a host value) and, every `eval_iter`-th episode, the number of tasks kept + the end of the trim.  Every
        statistic is a row of `_extra_buf`, filled by row assignments from device values."""
        import time
        t0 = time.perf_counter()
        E, Cn = self.num_envs, self.num_cylinders
        success = self.stats["success"]
        buf, row = self._extra_buf, self._extra_row
        split = self.num_unif < E
        flat = success.reshape(-1)
        rates = torch.stack([flat[self.num_unif:].mean() if split else flat.sum() * 0.0, flat[:self.num_unif].mean(), flat.mean()])
        if split:
            buf[row["success_buffer"]] = rates[0]
            buf[row["success_unif"]] = rates[1]
        else:
            buf[row["success_buffer"]] = 0.0
            buf[row["success_unif"]] = flat
        if self.global_gen_buffer:
            from . import sharding
            mean_success = sharding.global_mean(success)
        else:
            mean_success = float(rates[2])
        if mean_success > self.success_threshold:
            self.ratio_unif = 1.0
        self.gen_buffer.insert_weights(success)
        self.update_iter += 1
        if self.update_iter >= self.eval_iter:
            self.update_iter = 0
            counts, sums, n_kept = curriculum_update(self.gen_buffer, self.active_cylinders, Cn, self.R_min, self.R_max)
            r0, r1 = row["ratio_cylinders_0"], row["success_cylinders_0"]
            buf[r0:r0 + Cn + 1] = (counts / E).float().unsqueeze(1)
            buf[r1:r1 + Cn + 1] = torch.where(counts > 0, sums / counts.clamp(min=1.0), torch.zeros_like(sums)).float().unsqueeze(1)
            buf[row["add_history"]] = float(n_kept)
        buf[row["history_buffer"]] = float(len(self.gen_buffer))
        buf[row["ratio_unif"]] = self.ratio_unif
        self.generator_seconds += time.perf_counter() - t0

    def _clone_stats(self):
        """The base class's one-copy clone of the 24 task statistics + one copy of the generator's rows."""
        td = super()._clone_stats(skip=self._extra_row)
        extra = self._extra_buf.clone()
        for k, i in self._extra_row.items():
            td.set(k, extra[i].unsqueeze(-1))
        return td


HideAndSeek.REGISTRY["HideAndSeek_envgen"] = HideAndSeek_envgen
HideAndSeek.REGISTRY["hideandseek_envgen"] = HideAndSeek_envgen
