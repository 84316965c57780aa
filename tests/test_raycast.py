"""Ray-fan range sensor (extension): geometry of the oracle, and HIP == oracle."""
import numpy as np
import pytest

import hns_oracle as O
from hns_amd import config


def _setup(E=4, A=1, C=3):
    cfg = config.make_cfg({"num_agents": A, "cylinder": {"max_num": C, "min_num": 0, "obs_max_cylinder": 1}, "env": {"num_envs": E}})
    c = config.resolve_hns_cfg(cfg)
    arrs = O.alloc_buffers(c)
    arrs["drone_state"][..., 3] = 1.0                 # identity attitude: heading = +x
    arrs["drone_state"][..., 2] = 0.6
    arrs["cylinders"][..., 2] = -20.0                 # all inactive
    return c, arrs


def test_wall_and_cylinder_ranges():
    c, arrs = _setup()
    r = O.raycast(c, arrs, 8, 5.0)
    np.testing.assert_allclose(r[0, 0], 0.9, atol=1e-6)                  # centre of the arena: wall at 0.9 all around
    arrs["drone_state"][1, 0, 0] = 0.5                                   # off-centre along x
    r = O.raycast(c, arrs, 4, 5.0)
    np.testing.assert_allclose(r[1, 0], [0.4, np.sqrt(0.81 - 0.25), 1.4, np.sqrt(0.81 - 0.25)], atol=1e-6)
    arrs["cylinders"][2, 0] = [0.5, 0.0, 0.6]                            # cylinder dead ahead of the drone at the origin
    arrs["cylinders"][2, 1] = [0.0, -0.4, -20.0]                         # inactive one to the right: ignored
    r = O.raycast(c, arrs, 4, 5.0)
    np.testing.assert_allclose(r[2, 0], [0.4, 0.9, 0.9, 0.9], atol=1e-6)  # 0.5 - radius 0.1
    assert (O.raycast(c, arrs, 4, 0.3) <= 0.3).all()                     # clamp
    arrs["drone_state"][3, 0, :2] = [0.52, 0.0]
    arrs["cylinders"][3, 0] = [0.5, 0.0, 0.6]                            # origin inside a cylinder -> 0 range
    assert (O.raycast(c, arrs, 4, 5.0)[3, 0] == 0).all()
    # yaw 90 degrees: ray 0 now looks along +y
    arrs["drone_state"][1, 0, 3:7] = [np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)]
    r = O.raycast(c, arrs, 4, 5.0)
    np.testing.assert_allclose(r[1, 0], [np.sqrt(0.81 - 0.25), 1.4, np.sqrt(0.81 - 0.25), 0.4], atol=1e-5)


@pytest.mark.gpu
def test_raycast_hip_equals_oracle():
    import torch
    from hns_amd.env import HideAndSeek
    for A, C, N in ((3, 8, 16), (6, 16, 36), (1, 5, 7)):
        cfg = config.make_cfg({"num_agents": A, "cylinder": {"max_num": C, "min_num": min(4, C)}, "env": {"num_envs": 333}})
        env = HideAndSeek(cfg)
        env.set_seed(9)
        env.reset()
        for _ in range(6):
            env.step(env.rand_step_input())
        dev = env.raycast(N, 1.5).cpu().numpy()
        host = env.export_state()
        ref = O.raycast(env.hcfg, host, N, 1.5)
        np.testing.assert_array_equal(dev, ref)
        assert dev.shape == (333, A, N) and (dev >= 0).all() and (dev <= 1.5).all() and (dev < 1.5).any()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(16)))
def test_raycast_random_shapes_hip_equals_oracle(seed):
    """Seeded sweep: pursuers, cylinder slots (most of them inactive in some envs), ray counts, ranges, batch sizes, one or two evaders."""
    import torch
    from hns_amd.env import HideAndSeek
    r = np.random.RandomState(40 + seed)
    for _ in range(50):
        A, C = int(r.randint(1, 8)), int(r.randint(1, 17))
        task = {"num_agents": A, "num_targets": 2 if r.rand() < 0.25 else 1, "cylinder": {"max_num": C, "min_num": int(r.randint(0, C + 1)), "obs_max_cylinder": 1},
                "env": {"num_envs": int(r.choice([1, 63, 64, 200, 1000]))}}
        try:
            config.resolve_hns_cfg(config.make_cfg(task))
            break
        except ValueError:
            continue
    env = HideAndSeek(config.make_cfg(task))
    env.set_seed(seed)
    env.reset()
    E = env.num_envs
    for _ in range(int(r.randint(0, 8))):
        env.step(env.rand_step_input(torch.randn(E, A, 4, device=env.device)))
    N, rng_max = int(r.choice([1, 3, 8, 16, 33, 64])), float(r.choice([0.2, 1.5, 5.0]))
    dev = env.raycast(N, rng_max).cpu().numpy()
    ref = O.raycast(env.hcfg, env.export_state(), N, rng_max)
    np.testing.assert_array_equal(dev, ref, err_msg=f"seed {seed} {task} rays {N} range {rng_max}")
    assert dev.shape == (E, A, N) and (dev >= 0).all() and (dev <= rng_max).all()
