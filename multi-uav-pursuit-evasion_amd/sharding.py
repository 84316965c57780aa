"""Multi-GPU layout of the env batch: one process per GPU, contiguous env-index slices, and the
ONE collective the path needs.

Envs are independent units (every per-step interaction is intra-env), so rank r simply owns
envs [offset, offset+count) of every array and the step needs no data-path collective.  The
reference has two cross-env reductions, both outside the step proper:
  * advantage normalisation over the whole rollout (omni_drones/learning/mappo.py:391-396), and
  * `stats["success"].mean()` driving the evader-speed curriculum
    (omni_drones/envs/hide_and_seek/hideandseek.py:1012-1015).
Both are folded into a single all-gather of a tiny moment vector per rollout
(`[sum, sum_sq, count, success_sum, env_count]`), RCCL over xGMI on the GPUs
(`torch.distributed` backend "nccl"), gloo in the CPU tests.
"""
import torch
import torch.distributed as dist

MOMENT_DIM = 5


def env_shard(num_envs_total, world_size, rank):
    """Contiguous slice [offset, offset+count) of rank `rank`; the remainder goes to the low ranks."""
    base, rem = divmod(int(num_envs_total), int(world_size))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def local_moments(values, success=None):
    """[sum, sum of squares, count, success sum, env count] of this rank, fp64."""
    v = values.reshape(-1).double()
    out = torch.zeros(MOMENT_DIM, dtype=torch.float64, device=values.device)
    out[0], out[1], out[2] = v.sum(), (v * v).sum(), float(v.numel())
    if success is not None:
        s = success.reshape(-1).double()
        out[3], out[4] = s.sum(), float(s.numel())
    return out


def allgather_moments(local):
    """The rollout's single collective: every rank gets the [world, MOMENT_DIM] table."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local.unsqueeze(0)
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


def normalise_advantages(adv, success=None, eps=1e-7):
    """Data-parallel form of `(adv - adv.mean()) / adv.std().clip(1e-7)` (mappo.py:391-396).
    Returns (normalised advantages, global success rate or None)."""
    table = allgather_moments(local_moments(adv, success))
    mean, std = global_mean_std(table)
    rate = None
    if success is not None:
        tot = table.sum(0)
        rate = float(tot[3] / tot[4])
    return (adv - mean.to(adv.dtype)) / std.clamp(min=eps).to(adv.dtype), rate


class GlobalSuccessRate:
    """`env.success_rate_fn` for a shard of a data-parallel run: the success rate over the WHOLE batch, as the reference's
    `stats["success"].mean()` sees it (hideandseek.py:1012-1015), so that every shard raises the evader's speed at the
    same step.  The rate is the one gathered with the last rollout's moments (`update(table)`), i.e. one rollout old —
    episodes are 800 steps, rollouts 64; until the first rollout it falls back to the local shard."""

    def __init__(self):
        self.rate = None

    def update(self, table):
        tot = table.sum(0)
        if float(tot[4]) > 0:
            self.rate = float(tot[3] / tot[4])

    def __call__(self, env):
        return self.rate if self.rate is not None else float(env.stats["success"].mean())
