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


def test_top_down_raster_of_one_env():
    """`render(mode="rgb_array")`'s picture (env.rasterise_top_down; scripts/train.py:220-223,251-254 stacks such frames and transposes them to [N, 3, H, W]):
    uint8 [n, n, 3]; the arena disc, an active cylinder, the evader and a pursuer where they belong; an inactive slot (z = -20) is not drawn."""
    import numpy as np
    from hns_amd.env import rasterise_top_down
    A, NT, n, R = 2, 1, 128, 0.9
    pts = np.array([[0.5, 0.0, 0.6], [-0.5, 0.3, 1.2],          # pursuers (the second at the ceiling: brightest)
                    [0.0, -0.5, 0.6],                            # evader
                    [0.2, 0.4, 0.6], [-0.4, -0.4, -20.0]], np.float32)   # one active cylinder, one inactive slot
    img = rasterise_top_down(pts, A, NT, R, 1.2, 0.1, n)
    assert img.shape == (n, n, 3) and img.dtype == np.uint8

    def px(x, y):                                                # metres -> (row, column); row 0 is +y
        half = 1.15 * R
        return int((half - y) / (2 * half) * n), int((x + half) / (2 * half) * n)
    assert tuple(img[px(0.0, 0.0)]) == (58, 60, 66)              # inside the arena
    assert tuple(img[0, 0]) == (24, 24, 28)                      # outside
    assert tuple(img[px(0.2, 0.4)]) == (150, 150, 150)           # the active cylinder
    assert tuple(img[px(-0.4, -0.4)]) == (58, 60, 66)            # the inactive slot: plain arena
    assert tuple(img[px(0.0, -0.5)]) == (230, 60, 50)            # the evader
    p0, p1 = img[px(0.5, 0.0)], img[px(-0.5, 0.3)]
    assert p0[2] == 255 and p1[2] == 255 and p0[0] == 40 and p1[1] == 255 and p0[1] < p1[1]      # pursuers, brighter with height
    video = np.stack([img, img]).transpose(0, 3, 1, 2)          # what evaluate() hands to wandb.Video
    assert video.shape == (2, 3, n, n)
