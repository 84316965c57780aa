"""GPU parity of the device-side generator pieces (hns_fps, hns_perturb_tasks) through the C ABI:
bit-identical to the C oracle."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import hns_oracle as O
from hns_amd import abi, config
from hns_amd.env import HideAndSeek


def _fps_hip(lib, pts, k, start):
    dev = torch.device("cuda:0")
    p = torch.from_numpy(pts).to(dev).contiguous()
    out = torch.full((k,), -1, dtype=torch.int32, device=dev)
    scratch = torch.zeros(lib.hns_fps_scratch_bytes(), dtype=torch.uint8, device=dev)
    rc = lib.hns_fps(p.data_ptr(), pts.shape[0], pts.shape[1], k, start, out.data_ptr(), scratch.data_ptr(),
                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.hns_last_error()
    torch.cuda.synchronize()
    assert int(scratch[:8].view(torch.int64)[0]) == 0, "hns_fps gave up (a workgroup never showed up)"
    return out.cpu().numpy()


@pytest.mark.parametrize("n,d,k", [(1, 3, 1), (37, 5, 37), (300, 36, 300), (5000, 36, 800), (60000, 36, 400), (65536, 36, 64), (70001, 36, 150), (131072, 36, 48), (131073, 36, 16), (30000, 27, 300), (70000, 30, 64), (4096, 5, 100), (2048, 4, 64), (200000, 12, 40)])
def test_fps_matches_oracle(n, d, k):
    lib = abi.load_library()
    rng = np.random.default_rng(n + d)
    pts = rng.random((n, d), dtype=np.float32)
    if n > 10:
        pts[n // 3] = pts[1]                                   # exact duplicates: ties
    start = int(rng.integers(n))
    got = _fps_hip(lib, pts, k, start)
    ref = O.fps(pts, k, start)
    assert np.array_equal(got, ref)
    assert np.array_equal(got, _fps_hip(lib, pts, k, start))   # deterministic across launches


def test_fps_argument_errors():
    lib = abi.load_library()
    t = torch.zeros(16, device="cuda:0")
    assert lib.hns_fps(t.data_ptr(), 4, 4, 5, 0, t.data_ptr(), t.data_ptr(), None) == abi.HNS_ERR_INVALID_ARG     # k > n
    assert lib.hns_fps(t.data_ptr(), 4, 4, 2, 4, t.data_ptr(), t.data_ptr(), None) == abi.HNS_ERR_INVALID_ARG     # start >= n
    assert lib.hns_fps(None, 4, 4, 2, 0, t.data_ptr(), t.data_ptr(), None) == abi.HNS_ERR_INVALID_ARG


@pytest.mark.parametrize("A,Cn,expand", [(3, 5, 0), (3, 8, 1), (6, 16, 1), (1, 3, 0)])
def test_perturb_tasks_matches_oracle(A, Cn, expand):
    cfg = config.make_cfg({"num_agents": A, "cylinder": {"max_num": Cn, "min_num": min(2, Cn)}, "env": {"num_envs": 256}})
    env = HideAndSeek(cfg)
    env.set_seed(4)
    env.reset()
    b = env._bufs
    hist = torch.cat([b["drone_state"][..., :3].reshape(256, -1), b["target_pos"], b["cylinders"].reshape(256, -1)], dim=1).contiguous()
    n_tasks = 3000
    out = torch.zeros(n_tasks, hist.shape[1], device=env.device)
    rc = env._lib.hns_perturb_tasks(env._env, hist.data_ptr(), 256, out.data_ptr(), n_tasks, expand, C.c_float(0.1), C.c_uint64(99),
                                    env._stream())
    assert rc == 0, env._lib.hns_last_error()
    ref = O.perturb_tasks(env.hcfg, hist.cpu().numpy(), n_tasks, expand, 0.1, seed=99)
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("HNS_FUZZ_GEN_SEEDS", 24)))))
def test_fps_random_shapes_match_oracle(seed):
    """Seeded sweep: point counts around the kernels' hand-over sizes, every width up to 48 coordinates, duplicates, any start."""
    r = np.random.RandomState(300 + seed)
    n = int(r.choice([2, 63, 64, 65, 1000, 4097, 30000, 65535, 65537, 90000]))
    d = int(r.randint(1, 49))
    k = int(min(n, r.choice([1, 2, 17, 100, 300])))
    pts = r.rand(n, d).astype(np.float32)
    if seed % 3 == 0:
        pts[r.randint(0, n, size=n // 4)] = pts[r.randint(0, n)]      # duplicates: ties -> lower index, chosen points leave the pool
    start = int(r.randint(0, n))
    lib = abi.load_library()
    got = _fps_hip(lib, pts, k, start)
    assert np.array_equal(got, O.fps(pts, k, start)), f"seed {seed}: n={n} d={d} k={k} start={start}"
    assert len(set(got.tolist())) == k


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("HNS_FUZZ_GEN_SEEDS", 24)))))
def test_perturb_and_task_reset_random_shapes_match_oracle(seed):
    """Seeded sweep over pursuers / cylinder slots / cylinder size: hns_perturb_tasks and hns_reset_tasks bit-identical to the oracle."""
    r = np.random.RandomState(500 + seed)
    for _ in range(50):
        A, Cn = int(r.randint(1, 8)), int(r.randint(1, 17))
        task = {"num_agents": A, "cylinder": {"max_num": Cn, "min_num": int(r.randint(0, Cn + 1)), "obs_max_cylinder": int(r.randint(1, min(Cn, 6) + 1)),
                                              "size": float(r.choice([0.075, 0.1, 0.12]))},
                "env": {"num_envs": int(r.choice([64, 100, 256])), "max_episode_length": 20}}
        try:
            config.resolve_hns_cfg(config.make_cfg(task))
            break
        except ValueError:
            continue
    E = task["env"]["num_envs"]
    env = HideAndSeek(config.make_cfg(task))
    env.set_seed(seed)
    env.reset()
    b = env._bufs
    hist = torch.cat([b["drone_state"][..., :3].reshape(E, -1), b["target_pos"], b["cylinders"].reshape(E, -1)], dim=1).contiguous()
    n_tasks, expand, noise = int(r.choice([1, 77, 1000])), int(r.rand() < 0.5), float(r.choice([0.05, 0.1, 0.3]))
    out = torch.zeros(n_tasks, hist.shape[1], device=env.device)
    rc = env._lib.hns_perturb_tasks(env._env, hist.data_ptr(), E, out.data_ptr(), n_tasks, expand, C.c_float(noise), C.c_uint64(seed), env._stream())
    assert rc == 0, env._lib.hns_last_error()
    ref = O.perturb_tasks(env.hcfg, hist.cpu().numpy(), n_tasks, expand, noise, seed=seed)
    assert np.array_equal(out.cpu().numpy(), ref), f"seed {seed} {task}"
    # place the first E perturbed tasks (cycled) on the envs from `task_first` on: hns_reset_tasks against the oracle
    tasks = np.ascontiguousarray(ref[np.arange(E) % n_tasks])
    first = int(r.choice([0, E // 3, E]))
    tdev = torch.from_numpy(tasks).to(env.device)
    mask = (r.rand(E) < 0.7)
    mdev = torch.from_numpy(mask.astype(np.uint8)).to(env.device)
    host = env.export_state()
    epoch = env.reset_epoch
    rc = env._lib.hns_reset_tasks(env._env, C.c_void_p(mdev.data_ptr()), C.c_void_p(tdev.data_ptr()), C.c_int32(first), C.c_uint64(env.seed), env._stream())
    assert rc == 0, env._lib.hns_last_error()
    O.reset_tasks(env.hcfg, host, mask.astype(np.uint8), env.seed, epoch, tasks, first)
    st = env.export_state()
    for k in host:
        if k == "state_drones" and not host[k].size:
            continue
        np.testing.assert_array_equal(host[k], st[k], err_msg=f"seed {seed} {task}: {k}")
