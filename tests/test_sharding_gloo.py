"""N>1 path on CPU: world_size-2 gloo processes.  Each rank steps ITS contiguous env slice with the
oracle (the reset RNG is keyed by the global env index), and the only collective — the
per-rollout moment all-gather — reproduces the single-process statistics."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(E, offset):
    from hns_amd import config
    cfg = config.make_cfg({"num_agents": 3, "cylinder": {"max_num": 8, "min_num": 4}, "env": {"num_envs": E, "max_episode_length": 12}})
    return config.resolve_hns_cfg(cfg, env_index_offset=offset)


def _rollout(E, offset, actions, steps):
    import hns_oracle as O
    c = _cfg(E, offset)
    arrs = O.alloc_buffers(c)
    O.reset(c, arrs, None, 7, 0)
    rewards = []
    for t in range(steps):
        O.step(c, arrs, actions[t, offset:offset + E])
        rewards.append(arrs["reward"].copy())
    return arrs, np.stack(rewards)


def _worker(rank, world, port, E_total, steps, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import hns_amd  # noqa: F401
    from hns_amd import abi, sharding
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    offset, count = sharding.env_shard(E_total, world, rank)
    actions = np.random.default_rng(3).standard_normal((steps, E_total, 3, 4)).astype(np.float32)
    arrs, rew = _rollout(count, offset, actions, steps)
    adv = torch.from_numpy(rew)
    success = torch.from_numpy(arrs["stats"][abi.STAT_NAMES.index("success")])
    norm, rate = sharding.normalise_advantages(adv, success)
    q.put((rank, offset, count, arrs["drone_state"], arrs["stats"], rew, norm.numpy(), rate))
    dist.barrier()
    dist.destroy_process_group()


def test_env_shard_partition():
    from hns_amd import sharding
    for E, W in [(65536, 8), (10, 3), (7, 8), (524288, 8)]:
        spans = [sharding.env_shard(E, W, r) for r in range(W)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == E
        for (o0, c0), (o1, _) in zip(spans, spans[1:]):
            assert o0 + c0 == o1


@pytest.mark.timeout(300)
def test_two_rank_shards_equal_single_process():
    E_total, steps, world = 70, 14, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, E_total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    actions = np.random.default_rng(3).standard_normal((steps, E_total, 3, 4)).astype(np.float32)
    full, rew_full = _rollout(E_total, 0, actions, steps)
    from hns_amd import abi
    # shards are bit-identical slices of the single-process batch (state, stats, rewards)
    ds = np.concatenate([o[3] for o in outs])
    st = np.concatenate([o[4] for o in outs], axis=1)
    rew = np.concatenate([o[5] for o in outs], axis=1)
    np.testing.assert_array_equal(ds, full["drone_state"])
    np.testing.assert_array_equal(st, full["stats"])
    np.testing.assert_array_equal(rew, rew_full)
    # the one collective reproduces torch's global mean / unbiased std and the global success rate
    adv = torch.from_numpy(rew_full)
    ref = (adv - adv.mean()) / adv.std().clip(1e-7)
    got = np.concatenate([o[6] for o in outs], axis=1)
    np.testing.assert_allclose(got, ref.numpy(), rtol=2e-5, atol=2e-6)
    rate = float(full["stats"][abi.STAT_NAMES.index("success")].mean())
    assert abs(outs[0][7] - rate) < 1e-12 and abs(outs[1][7] - rate) < 1e-12
