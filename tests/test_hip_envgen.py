"""GPU parity of the device-side generator pieces (hns_fps, hns_perturb_tasks) through the C ABI:
bit-identical to the C oracle."""
import ctypes as C

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
