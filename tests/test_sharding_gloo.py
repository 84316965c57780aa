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


# ---- BASELINE config 5's job shape: 8 ranks, 6 pursuers / 2 evaders / 16 cylinders, an env count the ranks do not divide ---------------
def _job8_worker(rank, world, port, E_total, steps, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import hns_amd  # noqa: F401
    import hns_oracle as O
    from hns_amd import abi, config, sharding
    from hns_amd.envgen import GenBuffer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    offset, count = sharding.env_shard(E_total, world, rank)
    c = _cfg5(count, offset)
    arrs = O.alloc_buffers(c)
    O.reset(c, arrs, None, 7, 0)
    actions = np.random.default_rng(5).standard_normal((steps, E_total, 6, 4)).astype(np.float32)
    hook = sharding.GlobalSuccessRate()
    rews, rates = [], []
    for t in range(steps):
        O.step(c, arrs, actions[t, offset:offset + count])
        rews.append(arrs["reward"].copy())
        if (t + 1) % 4 == 0:                                   # a "rollout" of 4 steps: the job's one collective
            succ = torch.from_numpy(arrs["stats"][abi.STAT_NAMES.index("success")].copy())
            hook.update(sharding.allgather_moments(sharding.local_moments(torch.from_numpy(np.stack(rews[-4:])), succ)))
            rates.append(hook.rate)
    # one task history for the whole job: every rank contributes a different number of tasks (rank 3 none)
    gb = sharding.GlobalGenBuffer(GenBuffer(6, 16, seed=11, buffer_length=40, num_targets=2), num_envs_total=E_total)
    n = [5, 9, 2, 0, 7, 11, 3, 8][rank]
    gb.insert_history(np.random.default_rng(200 + rank).random((n, gb.task_dim)).astype(np.float32))
    q.put((rank, offset, count, arrs["drone_state"], arrs["target_pos"], arrs["stats"], np.stack(rews), rates,
           np.array(gb.inner._history_buffer, copy=True), gb.buffer_share(count, offset, 0.3)))
    dist.barrier()
    dist.destroy_process_group()


def _cfg5(E, offset):
    from hns_amd import config
    cfg = config.make_cfg({"num_agents": 6, "num_targets": 2, "cylinder": {"max_num": 16, "min_num": 16},
                           "env": {"num_envs": E, "max_episode_length": 9}})
    return config.resolve_hns_cfg(cfg, env_index_offset=offset)


@pytest.mark.timeout(600)
def test_eight_rank_job_of_config_5_equals_single_process():
    """BASELINE configs[4] as the driver will launch it — 8 ranks, contiguous env slices (env_shard(524288, 8) at full size; 77 envs here, so the
    slices are 10 and 9 envs), 6v2 / 16 cylinders — on gloo: the shards are bit-identical slices of the single-process batch, the per-rollout
    all-gather gives every rank the global success rate, and `GlobalGenBuffer` leaves all eight ranks with the history a single process
    builds from the concatenated contributions (one of them empty)."""
    import hns_oracle as O
    from hns_amd import abi, sharding
    from hns_amd.envgen import GenBuffer
    E_total, steps, world = 77, 12, 8
    spans = [sharding.env_shard(524288, 8, r) for r in range(8)]
    assert all(c == 65536 for _, c in spans) and [o for o, _ in spans] == [65536 * r for r in range(8)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_job8_worker, args=(r, world, port, E_total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=500) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [o[2] for o in outs] == [10, 10, 10, 10, 10, 9, 9, 9] and [o[1] for o in outs] == [0, 10, 20, 30, 40, 50, 59, 68]
    c = _cfg5(E_total, 0)
    full = O.alloc_buffers(c)
    O.reset(c, full, None, 7, 0)
    actions = np.random.default_rng(5).standard_normal((steps, E_total, 6, 4)).astype(np.float32)
    rews, rates = [], []
    for t in range(steps):
        O.step(c, full, actions[t])
        rews.append(full["reward"].copy())
        if (t + 1) % 4 == 0:
            rates.append(float(full["stats"][abi.STAT_NAMES.index("success")].astype(np.float64).mean()))
    np.testing.assert_array_equal(np.concatenate([o[3] for o in outs]), full["drone_state"])
    np.testing.assert_array_equal(np.concatenate([o[4] for o in outs]), full["target_pos"])
    np.testing.assert_array_equal(np.concatenate([o[5] for o in outs], axis=1), full["stats"])
    np.testing.assert_array_equal(np.concatenate([o[6] for o in outs], axis=1), np.stack(rews))
    for o in outs:                                            # the global rate, on every rank, at every rollout
        assert len(o[7]) == 3 and all(abs(a - b) < 1e-12 for a, b in zip(o[7], rates))
    single = GenBuffer(6, 16, seed=11, buffer_length=40, num_targets=2)
    single.insert_history(np.concatenate([np.random.default_rng(200 + r).random((n, single.task_dim)).astype(np.float32)
                                          for r, n in enumerate([5, 9, 2, 0, 7, 11, 3, 8])]))
    assert len(single._history_buffer) == 40                  # 45 contributed: trimmed once, on rank 0
    for o in outs:
        np.testing.assert_array_equal(o[8], single._history_buffer)
    assert sum(o[9] for o in outs) == min(40, int(E_total * 0.7))


# ---- one task history for the whole job (sharding.GlobalGenBuffer; reference semantics, hideandseek_envgen.py:209-233) ----------
def _genbuf_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import hns_amd  # noqa: F401
    from hns_amd import sharding
    from hns_amd.envgen import GenBuffer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gb = sharding.GlobalGenBuffer(GenBuffer(3, 5, seed=11, buffer_length=60), num_envs_total=200)
    rng = np.random.default_rng(100 + rank)
    hists = []
    for rnd in range(3):
        # rank 0 keeps 25 / 0 / 50 tasks, rank 1 keeps 40 / 30 / 45: an empty contribution and an overflow of the 60-entry history
        n = [[25, 0, 50], [40, 30, 45]][rank][rnd]
        gb.insert_history(rng.random((n, gb.task_dim)).astype(np.float32))
        hists.append(np.array(gb.inner._history_buffer, copy=True))
    share = gb.buffer_share(100, 100 * rank, 0.3)
    mean = sharding.global_mean(torch.full((100,), float(rank)))
    q.put((rank, hists, share, mean))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_global_gen_buffer_two_ranks():
    """Both ranks end every round with the SAME history, and it is the one a single process gets from the concatenated
    contributions (rank 0 does the appending and the farthest-point trim); the perturbed share of the next batch splits
    min(len(history), int(E_total (1 - ratio_unif))) over the ranks."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_genbuf_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from hns_amd.envgen import GenBuffer
    single = GenBuffer(3, 5, seed=11, buffer_length=60)            # rank 0's generator state (same seed)
    r0, r1 = np.random.default_rng(100), np.random.default_rng(101)
    for rnd, (n0, n1) in enumerate([(25, 40), (0, 30), (50, 45)]):
        both = np.concatenate([r0.random((n0, single.task_dim)).astype(np.float32), r1.random((n1, single.task_dim)).astype(np.float32)])
        single.insert_history(both)
        np.testing.assert_array_equal(outs[0][1][rnd], single._history_buffer)
        np.testing.assert_array_equal(outs[1][1][rnd], single._history_buffer)
    assert len(single._history_buffer) == 60
    assert outs[0][2] + outs[1][2] == min(60, int(200 * 0.7)) and abs(outs[0][2] - outs[1][2]) <= 1
    assert outs[0][3] == outs[1][3] == 0.5


def _envgen_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import hns_amd  # noqa: F401
    from hns_amd import config, sharding
    from hns_amd.envgen import HideAndSeek_envgen
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E, L = 192, 5
    cfg = config.make_cfg({"name": "HideAndSeek_envgen", "num_agents": 3, "cylinder": {"max_num": 5, "min_num": 0},
                           "use_particle_generator": 1, "ratio_unif": 0.3, "eval_iter": 2, "R_min": 0.0, "R_max": 1.0,
                           "global_gen_buffer": 1, "num_envs_total": E * world,
                           "env": {"num_envs": E, "max_episode_length": L}})
    env = HideAndSeek_envgen(cfg, headless=True, env_index_offset=rank * E)
    assert env.global_gen_buffer and isinstance(env.gen_buffer, sharding.GlobalGenBuffer)
    env.set_seed(3)
    td = env.reset()
    g = torch.Generator(device=env.device).manual_seed(rank)
    sizes, shares = [], []
    for ep in range(6):
        done = None
        while done is None or not bool(done.any()):
            td = env.step(env.rand_step_input(torch.randn(E, 3, 4, generator=g, device=env.device)))
            done = td[("next", "done")]
        r = env.rand_step_input()
        r.set("_reset", done.squeeze(-1))
        env.reset(r)
        sizes.append(len(env.gen_buffer))
        shares.append(E - env.num_unif)
    q.put((rank, sizes, shares, env.gen_buffer._history_buffer))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_envgen_global_history_two_ranks_one_gpu():
    """HideAndSeek_envgen with task.global_gen_buffer on two ranks (gloo; both on cuda:0): the device-resident buffers behind
    sharding.GlobalGenBuffer.  After every task batch the two ranks hold the same history, it grows by what BOTH kept, and the
    perturbed share of a batch is min(len(history), int(E_total 0.7)) split over the ranks."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_envgen_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=500) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, s0, sh0, h0), (_, s1, sh1, h1) = outs
    assert s0 == s1 and np.array_equal(h0, h1)                       # one history
    assert s0[1] == 2 * 192 and s0[3] == 4 * 192 and s0[5] == 6 * 192   # eval_iter = 2: every second episode both shards' 192 tasks enter
    assert sh0[0] == sh1[0] == 0                                      # empty history: uniform tasks only
    assert sh0[-1] + sh1[-1] == min(s0[-2], int(2 * 192 * 0.7))


@pytest.mark.gpu
def test_rccl_backend_runs_the_moment_collective_on_one_gpu(tmp_path):
    """The backend the multi-GPU run uses ("nccl" = RCCL on ROCm) with the one rank a single-GPU box allows: the process group
    initialises, the moment all-gather and the [sum, count] all-reduce run on device tensors, the normalised advantages equal torch's."""
    import subprocess
    import sys
    code = r"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import torch.distributed as dist
import hns_amd
from hns_amd import sharding
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29713")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
adv = torch.randn(64, 128, 3, 1, device="cuda:0")
suc = (torch.rand(128, 1, device="cuda:0") > 0.5).float()
table = sharding._allgather(sharding.local_moments(adv, suc))            # the collective itself, not the world_size == 1 shortcut
assert table.shape == (1, sharding.MOMENT_DIM) and table.is_cuda
mean, std = sharding.global_mean_std(table)
ref = (adv - adv.mean()) / adv.std().clip(1e-7)
got = (adv - mean.float()) / std.clamp(min=1e-7).float()
assert torch.allclose(got, ref, atol=1e-5), float((got - ref).abs().max())
acc = sharding._coll_tensor(torch.stack([suc.double().sum(), torch.tensor(float(suc.numel()), dtype=torch.float64, device="cuda:0")]))
dist.all_reduce(acc)
assert abs(float(acc[0] / acc[1]) - float(suc.mean())) < 1e-12
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
"""
    import os
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


@pytest.mark.gpu
def test_moments_in_one_launch_match_torch():
    """sharding.local_moments on device tensors = ONE launch of hns_moments (one workgroup, fixed order): the values of the torch form in fp64 to
    rounding, bit-identical from run to run, also for a pointer that is not 16-byte aligned and without a success row."""
    sys.path.insert(0, ROOT)
    import hns_amd  # noqa: F401
    from hns_amd import sharding
    g = torch.Generator(device="cuda").manual_seed(3)
    base = torch.randn(65536 * 3 + 5, generator=g, device="cuda") * 40 - 60
    suc = (torch.rand(65536, generator=g, device="cuda") < 0.02).float()
    for v, s in ((base[:65536 * 3].view(65536, 3), suc), (base[1:65536 * 3 + 4], suc[:777]), (base[:1000], None)):
        got = sharding.local_moments(v, s)
        again = sharding.local_moments(v, s)
        assert torch.equal(got, again)
        d = v.reshape(-1).double()
        ref = torch.tensor([float(d.sum()), float((d * d).sum()), float(d.numel()), float(s.double().sum()) if s is not None else 0.0,
                            float(s.numel()) if s is not None else 0.0], dtype=torch.float64)
        assert torch.allclose(got.cpu(), ref, rtol=1e-12, atol=1e-9), (got, ref)
    # the non-contiguous / non-fp32 forms still take the torch path
    got = sharding.local_moments(base[:2000:2], None)
    assert abs(float(got[0]) - float(base[:2000:2].double().sum())) < 1e-9
