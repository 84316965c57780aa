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
    ref = (adv - adv.mean()) / (adv.std() + 1e-8)               # learning/mappo.py:391-396
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
ref = (adv - adv.mean()) / (adv.std() + 1e-8)
got = (adv - mean.float()) / (std.float() + 1e-8)
assert torch.allclose(got, ref, atol=1e-5), float((got - ref).abs().max())
ret = torch.randn(64, 128, 3, 1, device="cuda:0") * 4 - 12
t2 = sharding._allgather(sharding.local_moments(adv, suc, ret))          # with ValueNorm1's batch moments riding along (SURVEY 8(e)(3))
assert torch.equal(t2[:, :5], table[:, :5])
bm, bsq = sharding.global_value_moments(t2)
assert abs(float(bm) - float(ret.double().mean())) < 1e-9 and abs(float(bsq) - float((ret.double() ** 2).mean())) < 1e-9
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
                            float(s.numel()) if s is not None else 0.0, 0.0, 0.0, 0.0], dtype=torch.float64)
        assert torch.allclose(got.cpu(), ref, rtol=1e-12, atol=1e-9), (got, ref)
        # the same launch with a second array (the rollout's returns): the first five entries do not move, the last three are its moments
        r = (v.reshape(-1) * 0.25 + 3.0).contiguous()[: max(1, v.numel() // 2)]
        both = sharding.local_moments(v, s, r)
        assert torch.equal(both[:5], got[:5]) and torch.equal(both, sharding.local_moments(v, s, r))
        rd = r.double()
        assert torch.allclose(both[5:].cpu(), torch.tensor([float(rd.sum()), float((rd * rd).sum()), float(rd.numel())], dtype=torch.float64), rtol=1e-12, atol=1e-9)
    # the non-contiguous / non-fp32 forms still take the torch path
    got = sharding.local_moments(base[:2000:2], None)
    assert abs(float(got[0]) - float(base[:2000:2].double().sum())) < 1e-9


# ---- the collective's consumers against the reference's learner ----------------------------------------------------------------
class _ValueNorm1:
    """The state and the two read-outs of the reference's ValueNorm1 (learning/utils/valuenorm.py:45-106) that `valuenorm1_update` works on — the class itself
    cannot travel to the GPU box; g_learner_moments.npz holds what the reference's instance held after each update."""

    def __init__(self, beta, epsilon=1e-5):
        self.beta, self.epsilon = beta, epsilon
        self.running_mean, self.running_mean_sq, self.debiasing_term = torch.zeros(1), torch.zeros(1), torch.tensor(0.0)

    def _mean_var(self):
        d = self.debiasing_term.clamp(min=self.epsilon)
        mean, mean_sq = self.running_mean / d, self.running_mean_sq / d
        return mean, (mean_sq - mean ** 2).clamp(min=1e-2)

    def normalize(self, x):
        mean, var = self._mean_var()
        return (x - mean) / torch.sqrt(var)

    def denormalize(self, x):
        mean, var = self._mean_var()
        return x * torch.sqrt(var) + mean


def _check_learner_golden(g, r, adv_n, ret_n, vn, what):
    # fp32 statements on [E,T,A,1] in the reference against fp64 moments cast to fp32 here: a few ulp of the mean / std
    np.testing.assert_allclose(adv_n, g[f"r{r}_adv_normalised"], rtol=3e-6, atol=3e-6, err_msg=what)
    np.testing.assert_allclose(vn.running_mean.numpy(), g[f"r{r}_running_mean"], rtol=2e-6, atol=1e-7, err_msg=what)
    np.testing.assert_allclose(vn.running_mean_sq.numpy(), g[f"r{r}_running_mean_sq"], rtol=2e-6, atol=1e-7, err_msg=what)
    np.testing.assert_allclose(float(vn.debiasing_term), float(g[f"r{r}_debiasing_term"]), rtol=1e-6, err_msg=what)
    np.testing.assert_allclose(ret_n, g[f"r{r}_ret_normalised"], rtol=2e-5, atol=2e-5, err_msg=what)
    np.testing.assert_allclose(vn.denormalize(torch.linspace(-2, 2, 9).unsqueeze(-1)).numpy(), g[f"r{r}_denormalised_probe"], rtol=2e-5, atol=2e-5, err_msg=what)


def test_learner_moments_single_process_lands_on_the_reference(golden):
    """world_size 1: `normalise_advantages` + `valuenorm1_update` on the whole tensors = mappo.py:391-402 / valuenorm.py:83-98 executed as written."""
    from hns_amd import sharding
    g = golden("g_learner_moments")
    E, T, A, R = (int(x) for x in g["meta"])
    vn = _ValueNorm1(float(g["beta"]))
    for r in range(R):
        adv, ret = torch.from_numpy(g[f"r{r}_adv"]), torch.from_numpy(g[f"r{r}_ret"])
        adv_n, _ = sharding.normalise_advantages(adv, returns=ret, value_normalizer=vn)
        _check_learner_golden(g, r, adv_n.numpy(), vn.normalize(ret).numpy(), vn, f"rollout {r}")
    # the single-agent learner's form stays available and differs only in the guard (learning/_ppo.py:176)
    a = torch.from_numpy(g["r0_adv"])
    np.testing.assert_allclose(sharding.normalise_advantages(a, form="ppo")[0].numpy(), ((a - a.mean()) / a.std().clip(1e-7)).numpy(), rtol=3e-6, atol=3e-6)
    with pytest.raises(ValueError):
        sharding.normalise_advantages(a, form="tianshou")


def _learner_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import hns_amd  # noqa: F401
    from hns_amd import sharding
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g_learner_moments.npz")))
    E, T, A, R = (int(x) for x in g["meta"])
    off, cnt = sharding.env_shard(E, world, rank)
    vn = _ValueNorm1(float(g["beta"]))
    outs = []
    for r in range(R):
        adv, ret = torch.from_numpy(g[f"r{r}_adv"][off:off + cnt]).contiguous(), torch.from_numpy(g[f"r{r}_ret"][off:off + cnt]).contiguous()
        adv_n, _ = sharding.normalise_advantages(adv, returns=ret, value_normalizer=vn)
        outs.append((adv_n.numpy(), vn.normalize(ret).numpy(), vn.running_mean.numpy().copy(), vn.running_mean_sq.numpy().copy(), vn.debiasing_term.numpy().copy()))   # numpy: torch tensors travel as shared-memory handles the exiting worker takes with it
    q.put((rank, off, cnt, outs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 8])
def test_learner_moments_sharded_lands_on_the_reference(world, golden):
    """The same tensors split over 2 / 8 ranks by contiguous env slices: every rank's normalised advantages are the slice of the reference's, and every
    rank holds the SAME value normaliser a single process would hold — from the one all-gather."""
    g = golden("g_learner_moments")
    E, T, A, R = (int(x) for x in g["meta"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 17 * world) % 2000
    procs = [ctx.Process(target=_learner_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=500) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(R):
        adv_n = np.concatenate([o[3][r][0] for o in outs])
        ret_n = np.concatenate([o[3][r][1] for o in outs])
        for o in outs[1:]:                                    # one normaliser: bit-identical on every rank
            assert all(np.array_equal(x, y) for x, y in zip(o[3][r][2:], outs[0][3][r][2:]))
        vn = _ValueNorm1(float(g["beta"]))
        vn.running_mean, vn.running_mean_sq, vn.debiasing_term = (torch.from_numpy(np.asarray(x)) for x in outs[0][3][r][2:])
        _check_learner_golden(g, r, adv_n, ret_n, vn, f"world {world} rollout {r}")


@pytest.mark.gpu
def test_learner_moments_on_device_land_on_the_reference(golden):
    """The device path: one launch of hns_rollout_moments per rollout feeds both consumers (through the C ABI), against the same fixture."""
    from hns_amd import sharding
    g = golden("g_learner_moments")
    E, T, A, R = (int(x) for x in g["meta"])
    vn = _ValueNorm1(float(g["beta"]))
    vn.running_mean, vn.running_mean_sq, vn.debiasing_term = vn.running_mean.cuda(), vn.running_mean_sq.cuda(), vn.debiasing_term.cuda()
    for r in range(R):
        adv, ret = torch.from_numpy(g[f"r{r}_adv"]).cuda(), torch.from_numpy(g[f"r{r}_ret"]).cuda()
        assert adv.numel() <= sharding.HNS_MOMENTS_MAX
        adv_n, _ = sharding.normalise_advantages(adv, returns=ret, value_normalizer=vn)
        host = _ValueNorm1(float(g["beta"]))
        host.running_mean, host.running_mean_sq, host.debiasing_term = vn.running_mean.cpu(), vn.running_mean_sq.cpu(), vn.debiasing_term.cpu()
        _check_learner_golden(g, r, adv_n.cpu().numpy(), vn.normalize(ret).cpu().numpy(), host, f"device rollout {r}")
