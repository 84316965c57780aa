"""GPU: Hover (BASELINE config 1: 1 Crazyflie, 16 envs) through the C ABI — plumbing (shapes,
reset/step/set_seed) and bit-exact parity with the oracle, incl. the episode boundary."""
import numpy as np
import pytest
import torch

import hns_oracle as O
from hns_amd import abi, config

pytestmark = pytest.mark.gpu


def test_hover_plumbing_and_parity():
    from hns_amd.hover import Hover
    from hns_amd.env import HideAndSeek
    cfg = config.make_hover_cfg({"env": {"num_envs": 16, "max_episode_length": 12}})
    env = HideAndSeek.REGISTRY[cfg.task.name](cfg, headless=True)
    assert isinstance(env, Hover)
    env.set_seed(0)
    td = env.reset()
    assert td[("agents", "observation")].shape == (16, 1, 20) and td["done"].shape == (16, 1)
    host = O.alloc_hover_buffers(env.hcfg)
    O.hover_reset(env.hcfg, env.hover_cfg, host, None, 0, 0)
    dev = env.export_state()
    for k in host:
        np.testing.assert_array_equal(host[k], dev[k], err_msg=f"reset: {k}")
    g = torch.Generator().manual_seed(0)
    for t in range(20):
        action = torch.randn(16, 1, 4, generator=g)
        td = env.step(env.rand_step_input(action.to(env.device)))
        O.hover_step(env.hcfg, env.hover_cfg, host, action.numpy())
        assert td[("next", "agents", "reward")].shape == (16, 1, 1)
        if t == 11:
            assert bool(td[("next", "done")].all())
            env.reset()
            O.hover_reset(env.hcfg, env.hover_cfg, host, None, 0, 1)
    dev = env.export_state()
    for k in host:
        np.testing.assert_array_equal(host[k], dev[k], err_msg=f"step: {k}")
    assert set(env.stats.keys()) == set(abi.HOVER_STAT_NAMES)


def test_hover_matches_reference_golden(golden):
    from hns_amd.hover import Hover
    g = golden("g_hover")
    E, T, max_len = (int(x) for x in g["meta"])
    env = Hover(config.make_hover_cfg({"env": {"num_envs": E, "max_episode_length": max_len}}))
    env.reset()
    st = env.export_state()
    for t in range(T):
        if t == 0:
            pos, rot, vel, thr, prog, stats, acc = (g["init_" + k] for k in ("pos", "rot", "vel", "throttle", "progress", "stats", "acc"))
            prev = np.zeros((E, 1, 4), np.float32)
            integ = last = np.zeros((E, 1, 3), np.float32)
        else:
            pos, rot, vel, thr, prog, stats, acc = (g[k][t - 1] for k in ("pos", "rot", "vel", "throttle", "progress", "stats", "acc"))
            prev, integ, last = g["prev_action"][t - 1], g["integ"][t - 1], g["last"][t - 1]
        st["drone_state"][..., 0:3], st["drone_state"][..., 3:7], st["drone_state"][..., 7:13] = pos, rot, vel
        st["throttle"][:], st["prev_action"][:], st["progress"][:] = thr, prev, prog
        st["stats"][:], st["acc"][:] = stats.T, acc.T
        st["pid_integ"][..., :3], st["pid_last_rate"][..., :3] = integ, last
        env.import_state(st)
        env.step(env.rand_step_input(torch.as_tensor(g["action"][t]).to(env.device)))
        out = env.export_state()
        np.testing.assert_allclose(out["drone_state"][..., 0:3], g["pos"][t], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(out["obs"], g["obs"][t], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(out["reward"], g["reward"][t][..., 0], rtol=1e-5, atol=2e-6)
        assert (out["done"].astype(bool) == g["done"][t][:, 0]).all()


@pytest.mark.parametrize("seed", list(range(12)))
def test_hover_random_batches_bit_exact(seed):
    """Seeded sweep: batch sizes (ragged against the 64-thread workgroups), episode lengths, action scales; masked resets at the
    episode boundary; every buffer bit for bit against the oracle."""
    from hns_amd.env import HideAndSeek
    r = np.random.RandomState(70 + seed)
    E, L = int(r.choice([1, 16, 63, 65, 300, 1024])), int(r.randint(4, 15))
    cfg = config.make_hover_cfg({"env": {"num_envs": E, "max_episode_length": L}})
    env = HideAndSeek.REGISTRY[cfg.task.name](cfg, headless=True)
    env.set_seed(seed)
    env.reset()
    host = O.alloc_hover_buffers(env.hcfg)
    O.hover_reset(env.hcfg, env.hover_cfg, host, None, seed, 0)
    g = torch.Generator().manual_seed(seed)
    epoch = 1
    for t in range(3 * L):
        action = torch.randn(E, 1, 4, generator=g) * float(r.choice([0.3, 1.0, 3.0]))
        td = env.step(env.rand_step_input(action.to(env.device)))
        O.hover_step(env.hcfg, env.hover_cfg, host, action.numpy())
        if host["done"].any():
            mask = host["done"].copy()
            rtd = env.rand_step_input()
            rtd.set("_reset", torch.as_tensor(mask.astype(bool), device=env.device))
            env.reset(rtd)
            O.hover_reset(env.hcfg, env.hover_cfg, host, mask, seed, epoch)
            epoch += 1
        if t % L == L - 1 or t == 3 * L - 1:
            dev = env.export_state()
            for k in host:
                np.testing.assert_array_equal(host[k], dev[k], err_msg=f"seed {seed} E={E} L={L} step {t}: {k}")
