"""Host-side pieces that need no GPU: the TensorDict/spec shim and the config resolver's
reference-form constants."""
import torch

from hns_amd import config
from hns_amd.tensordict_shim import CompositeSpec, TensorDict, TensorSpec, USING_REAL_TENSORDICT


def test_tensordict_shim_surface():
    if USING_REAL_TENSORDICT:
        return
    td = TensorDict({"agents": {"action": torch.zeros(4, 3, 4)}, "done": torch.zeros(4, 1, dtype=torch.bool)}, [4])
    assert td[("agents", "action")].shape == (4, 3, 4)
    td.set(("info", "prev_action"), torch.ones(4, 3, 4))
    assert ("info", "prev_action") in td.keys(True, True) and "info" in td.keys()
    c = td.clone()
    c[("info", "prev_action")].zero_()
    assert td[("info", "prev_action")].sum() == 48
    sub = td[1:3]
    assert sub.batch_size == torch.Size([2]) and sub[("agents", "action")].shape == (2, 3, 4)
    assert td.get("missing") is None
    assert "done" not in td.exclude("done").keys()
    td.update({"agents": {"reward": torch.zeros(4, 3, 1)}})
    assert set(td["agents"].keys()) == {"action", "reward"}


def test_spec_tree():
    spec = CompositeSpec({"agents": CompositeSpec({"action": TensorSpec((3, 4), low=-1.0, high=1.0)})}).expand(8)
    assert spec[("agents", "action")].shape == (8, 3, 4)
    assert spec.zero()[("agents", "action")].shape == (8, 3, 4)
    r = spec.rand()[("agents", "action")]
    assert float(r.min()) >= -1 and float(r.max()) <= 1


def test_reference_form_constants():
    c = config.resolve_hns_cfg(config.make_cfg())
    # torch evaluates `dt / tau` as tau.reciprocal()*dt (rotor_group.py:61): 0.39999998, not 0.4
    assert abs(c.tau_up - 0.39999998) < 1e-8 and c.tau_up != 0.4
    assert abs(c.hover_throttle - (0.0321 * 9.81 / (4 * c.kf[0])) ** 0.5) < 1e-6
    assert abs(c.max_lin_vel - 1.0) < 2e-6 and c.max_lin_vel < 1.0
    assert abs(c.drone_xy_hi[0] - (0.9 / 2 ** 0.5 - 0.1)) < 1e-6
    e = config.resolve_hns_cfg(config.make_cfg({"use_eval": 1}))
    assert e.init_mode == 1 and list(e.rpy_hi) == [0.0, 0.0, 0.0]


def test_bench_algorithmic_bytes_match_the_survey_figures():
    """B_env(A,C,k,S) — 1 533 B (3v1, 8 cylinders), 1 497 B (5 slots), 3 045 B (6 pursuers, 16 cylinders)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.algorithmic_bytes_per_env(3, 8, 3) == 1533
    assert bench.algorithmic_bytes_per_env(3, 5, 3) == 1497
    assert bench.algorithmic_bytes_per_env(6, 16, 3) == 3045
    assert bench.algorithmic_bytes_per_env(6, 16, 3, NT=2) == 3045 + 12 + 24 + 16 * 6
