"""Multi-GPU layout of the env batch: one process per GPU, contiguous env-index slices, and the
ONE collective the path needs.

Envs are independent units (every per-step interaction is intra-env), so rank r simply owns
envs [offset, offset+count) of every array and the step needs no data-path collective.  The
reference has three cross-env reductions, all outside the step proper:
  * advantage normalisation over the whole rollout (omni_drones/learning/mappo.py:391-396),
  * `stats["success"].mean()` driving the evader-speed curriculum
    (omni_drones/envs/hide_and_seek/hideandseek.py:1012-1015), and
  * ValueNorm1's batch moments of the rollout's returns (learning/utils/valuenorm.py:83-91 via mappo.py:398-399).
All are folded into a single all-gather of a tiny moment vector per rollout (MOMENT_DIM fp64 values, below), RCCL over xGMI
on the GPUs (`torch.distributed` backend "nccl"), gloo in the CPU tests.  The consumers reproduce what the reference's learner
computes on the whole batch (tests/golden/g_learner_moments.npz: those statements executed as written).
"""
import torch
import torch.distributed as dist

# [sum adv, sum adv^2, n_adv, sum success, n_envs, sum returns, sum returns^2, n_returns] — one row per rank
MOMENT_DIM = 8
HNS_MOMENTS_MAX = 1 << 18          # elements one workgroup of hns_moments reads in ~10 us (1 MB); larger tensors go to torch's reductions


def env_shard(num_envs_total, world_size, rank):
    """Contiguous slice [offset, offset+count) of rank `rank`; the remainder goes to the low ranks."""
    base, rem = divmod(int(num_envs_total), int(world_size))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def _small_dev_f32(t):
    return t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() <= HNS_MOMENTS_MAX)


def local_moments(values, success=None, returns=None):
    """This rank's row of the moment table (MOMENT_DIM fp64 values; entries of an absent input are 0).  Small contiguous fp32 device tensors go
    through ONE launch of the library (hns_moments / hns_rollout_moments: one workgroup, fixed summation order) instead of a dozen small torch kernels in
    front of the collective; above HNS_MOMENTS_MAX elements (a whole rollout's advantages: 65 536 x 64 x 3 = 12.6 M values) one workgroup would read
    ~50 MB from one compute unit, so torch's chip-wide reductions take over."""
    if values.is_cuda and _small_dev_f32(values) and _small_dev_f32(success) and _small_dev_f32(returns):
        import ctypes as C
        from . import abi
        lib = abi.load_library()
        out = torch.zeros(MOMENT_DIM, dtype=torch.float64, device=values.device)
        sp, sn = (success.data_ptr(), success.numel()) if success is not None else (None, 0)
        st = C.c_void_p(torch.cuda.current_stream(values.device).cuda_stream)
        with torch.cuda.device(values.device):
            if returns is None:
                rc = lib.hns_moments(values.data_ptr(), values.numel(), sp, sn, out.data_ptr(), st)
            else:
                rc = lib.hns_rollout_moments(values.data_ptr(), values.numel(), sp, sn, returns.data_ptr(), returns.numel(), out.data_ptr(), st)
        if rc != 0:
            raise RuntimeError(f"hns_moments failed ({rc}): {lib.hns_last_error().decode()}")
        return out
    v = values.reshape(-1).double()
    out = torch.zeros(MOMENT_DIM, dtype=torch.float64, device=values.device)
    out[0], out[1], out[2] = v.sum(), (v * v).sum(), float(v.numel())
    if success is not None:
        s = success.reshape(-1).double()
        out[3], out[4] = s.sum(), float(s.numel())
    if returns is not None:
        r = returns.reshape(-1).double()
        out[5], out[6], out[7] = r.sum(), (r * r).sum(), float(r.numel())
    return out


def allgather_moments(local):
    """The rollout's single collective: every rank gets the [world, MOMENT_DIM] table."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local.unsqueeze(0)
    return _allgather(local)


def _allgather(local):
    """The collective itself (also called directly by the one-rank RCCL test on a single GPU)."""
    src = local.contiguous() if dist.get_backend() == "nccl" else local.cpu()   # gloo (CPU tests) gathers host tensors
    parts = [torch.empty_like(src) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, src)
    return torch.stack(parts).to(local.device)


def global_mean_std(table):
    """mean and UNBIASED std (torch.std default, as mappo.py:393-394 uses) from gathered moments."""
    tot = table.sum(0)
    n = tot[2]
    mean = tot[0] / n
    var = (tot[1] - n * mean * mean) / (n - 1)
    return mean, torch.sqrt(torch.clamp(var, min=0.0))


def global_value_moments(table):
    """ValueNorm1.update's two batch moments over the WHOLE sharded batch (valuenorm.py:86-87: `input_vector.mean(dim)`, `(input_vector**2).mean(dim)`
    with input_shape (1,), i.e. over every leading dimension) from the gathered table, fp64."""
    tot = table.sum(0)
    return tot[5] / tot[7], tot[6] / tot[7]


def valuenorm1_update(value_normalizer, table):
    """`self.value_normalizer.update(tensordict["returns"])` (mappo.py:399) for a shard of a data-parallel run: the reference's ValueNorm1 instance
    (running_mean, running_mean_sq, debiasing_term, beta; valuenorm.py:45-91) updated with the batch moments of ALL ranks' returns, so every rank holds
    the normaliser a single process would hold.  Lines :88-91 in meaning, with the two batch means taken from the gathered table."""
    batch_mean, batch_sq_mean = global_value_moments(table)
    vn, w = value_normalizer, float(value_normalizer.beta)
    with torch.no_grad():
        vn.running_mean.mul_(w).add_(batch_mean.to(vn.running_mean) * (1.0 - w))
        vn.running_mean_sq.mul_(w).add_(batch_sq_mean.to(vn.running_mean_sq) * (1.0 - w))
        vn.debiasing_term.mul_(w).add_(1.0 * (1.0 - w))


def normalise_advantages(adv, success=None, eps=None, form="mappo", returns=None, value_normalizer=None):
    """Data-parallel form of the learner's advantage normalisation over the whole rollout, from ONE all-gather of the moment table.

    form="mappo" (default — the learner on this path): `(adv - mean) / (std + 1e-8)` as MAPPOPolicy.train_op has it (learning/mappo.py:391-396;
    `torch.std` = unbiased).  form="ppo": `(adv - mean) / std.clip(1e-7)`, the single-agent learner's (learning/_ppo.py:176).  `eps` overrides the constant.
    With `returns` the table also carries ValueNorm1's batch moments; a `value_normalizer` (the reference's ValueNorm1 instance) is
    then updated in place with the global moments (`valuenorm1_update`), as mappo.py:398-399 does right behind the normalisation.
    Returns (normalised advantages, global success rate or None)."""
    if form not in ("mappo", "ppo"):
        raise ValueError("form must be 'mappo' (learning/mappo.py:391-396) or 'ppo' (learning/_ppo.py:176)")
    table = allgather_moments(local_moments(adv, success, returns))
    mean, std = global_mean_std(table)
    rate = None
    if success is not None:
        tot = table.sum(0)
        rate = float(tot[3] / tot[4])
    if value_normalizer is not None:
        if returns is None:
            raise ValueError("a value_normalizer needs the rollout's returns")
        valuenorm1_update(value_normalizer, table)
    mean, std = mean.to(adv.dtype), std.to(adv.dtype)
    if form == "mappo":
        return (adv - mean) / (std + (1e-8 if eps is None else eps)), rate
    return (adv - mean) / std.clamp(min=1e-7 if eps is None else eps), rate


def _coll_tensor(t):
    """A tensor the active backend can move: device tensors for RCCL, host tensors for gloo."""
    return t.contiguous() if dist.get_backend() == "nccl" else t.cpu().contiguous()


def global_mean(values):
    """Mean over the WHOLE sharded batch (one all-reduce of [sum, count])."""
    v = values.reshape(-1).double()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(v.mean())
    acc = _coll_tensor(torch.stack([v.sum(), torch.tensor(float(v.numel()), dtype=torch.float64, device=v.device)]))
    dist.all_reduce(acc)
    return float(acc[0] / acc[1])


class GlobalGenBuffer:
    """ONE task history for a data-parallel run, as the reference's single process has it (hideandseek_envgen.py:209-233):
    rank 0 owns the 5000-entry history.  When a task batch has been evaluated every rank hands the tasks it keeps to rank 0
    (one gather, variable length, padded), rank 0 appends them and runs the farthest-point trim once, and the trimmed history
    (<= 5000 x task_dim fp32 = 0.7 MB) goes back to every rank in one broadcast; each rank then perturbs ITS share of the next
    batch from that common history with its own Philox stream.  Wraps a GenBuffer (numpy, CPU tests) or a DeviceGenBuffer.
    Traffic per task batch over xGMI: <= 65 536 x 144 B = 9.4 MB into rank 0 per rank in the worst case (every task kept), 0.7 MB out.
    Without it (the default) every rank keeps its own 5000-entry history: W histories of the shard's tasks — cheaper, no exchange,
    but 8 x 5000 remembered tasks instead of 5000 and no task crosses shards."""

    def __init__(self, inner, num_envs_total=None):
        object.__setattr__(self, "inner", inner)
        object.__setattr__(self, "num_envs_total", num_envs_total)

    def __getattr__(self, name):
        return getattr(self.inner, name)

    def __setattr__(self, name, value):
        setattr(self.inner, name, value)

    def __len__(self):
        return self._n()

    def _n(self):
        return int(self.inner._history.shape[0]) if hasattr(self.inner, "_history") else int(self.inner._history_buffer.shape[0])

    def _history_tensor(self):
        h = getattr(self.inner, "_history", None)
        return h if h is not None else torch.as_tensor(self.inner._history_buffer)

    def insert_history(self, states):
        world, rank = dist.get_world_size(), dist.get_rank()
        D = self.inner.task_dim
        dev = getattr(self.inner, "_history", torch.zeros(0)).device if hasattr(self.inner, "_history") else torch.device("cpu")
        kept = torch.as_tensor(states, dtype=torch.float32).reshape(-1, D).to(dev)
        # 1. how many tasks every rank keeps, then the tasks themselves, padded to the longest (gather to rank 0)
        n_local = _coll_tensor(torch.tensor([kept.shape[0]], dtype=torch.int64, device=dev))
        counts = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(counts, n_local)
        counts = [int(c.item()) for c in counts]
        longest = max(counts)
        n_hist = _coll_tensor(torch.zeros(1, dtype=torch.int64, device=dev))
        if longest > 0:
            pad = torch.zeros(longest, D, dtype=torch.float32, device=dev)
            pad[:kept.shape[0]] = kept
            pad = _coll_tensor(pad)
            parts = [torch.zeros_like(pad) for _ in range(world)] if rank == 0 else None
            dist.gather(pad, parts, dst=0)
            if rank == 0:
                allkept = torch.cat([p[:c] for p, c in zip(parts, counts)]).to(dev)
                self.inner.insert_history(allkept if hasattr(self.inner, "_history") else allkept.cpu().numpy())
        # 2. the trimmed history back to everybody
        if rank == 0:
            n_hist[0] = self._n()
        dist.broadcast(n_hist, src=0)
        n = int(n_hist.item())
        hist = _coll_tensor(self._history_tensor().to(dev).float()) if rank == 0 else _coll_tensor(torch.zeros(n, D, dtype=torch.float32, device=dev))
        if n > 0:
            dist.broadcast(hist, src=0)
            if rank != 0:
                self.inner.init_history(hist.cpu().numpy())

    def buffer_share(self, num_envs_local, env_offset, ratio_unif):
        """How many of this rank's envs take a perturbed history task: the reference's min(len(history), int(E (1 - ratio_unif)))
        (hideandseek_envgen.py:881-883) with E the WHOLE batch, split over the ranks in proportion to their env ranges."""
        total = self.num_envs_total or num_envs_local * dist.get_world_size()
        nb = min(self._n(), int(total * (1 - ratio_unif)))
        lo, hi = env_offset * nb // total, (env_offset + num_envs_local) * nb // total
        return hi - lo


class GlobalSuccessRate:
    """`env.success_rate_fn` for a shard of a data-parallel run: the success rate over the WHOLE batch, as the reference's
    `stats["success"].mean()` sees it (hideandseek.py:1012-1015), so that every shard raises the evader's speed at the
    same step.  The rate is the one gathered with the last rollout's moments (`update(table)`), i.e. one rollout old —
    episodes are 800 steps, rollouts 64; until the first rollout it falls back to the local shard."""

    def __init__(self):
        self._table = None
        self._rate = None
        self._fresh = False

    def update(self, table):
        """Keep the gathered moments; nothing is read back here (a `.item()` per rollout would drain the launch queue every 64 steps).
        The rate is read back ONCE per table, by the first `rate` access after this call — `env._step` asks on every step once the
        curriculum is active and an episode length has passed, and must not pay a host synchronisation each time."""
        self._table = table
        self._fresh = False

    @property
    def rate(self):
        if self._table is None:
            return None
        if not self._fresh:
            tot = self._table.sum(0)[3:5].tolist()                  # one read-back: [sum of success, count]
            self._rate = tot[0] / tot[1] if tot[1] > 0 else None
            self._fresh = True
        return self._rate

    def __call__(self, env):
        r = self.rate
        return r if r is not None else float(env.stats["success"].mean())
