"""what `env.reset()` / `env.step()` hand to the caller, key by key, against the manifest that
tests/golden/make_golden.py::gen_manifest wrote by executing the REFERENCE's own `_set_specs`, `IsaacEnv._reset` and
`IsaacEnv._step` (hideandseek.py:327-433, isaac_env.py:210-240) on shims — names, shapes, dtypes, for use_TP_net 0 and 1."""
import json
import os

import pytest
import torch

from hns_amd import config
from hns_amd.tensordict_shim import spec_tree

HERE = os.path.dirname(os.path.abspath(__file__))
MANIFEST = json.load(open(os.path.join(HERE, "golden", "g_manifest.json")))
ME = MANIFEST["meta"]["num_envs"]


def leaves(tree, prefix=()):
    for k, v in tree.items():
        if isinstance(v, dict) and not ("shape" in v and "dtype" in v):
            yield from leaves(v, prefix + (k,))
        else:
            yield prefix + (k,), v


def td_leaves(td, prefix=()):
    for k in td.keys():
        v = td[k]
        if hasattr(v, "keys"):
            yield from td_leaves(v, prefix + (k,))
        else:
            yield prefix + (k,), v


def check_tree(want, got_td, E, allowed_extra=()):
    got = dict(td_leaves(got_td))
    for key, rec in leaves(want):
        assert key in got, f"missing key {key}"
        shape = [E if (i == 0 and d == ME) else d for i, d in enumerate(rec["shape"])]
        assert list(got[key].shape) == shape, f"{key}: shape {list(got[key].shape)} != reference {shape}"
        assert str(got[key].dtype).replace("torch.", "") == rec["dtype"], f"{key}: dtype {got[key].dtype} != {rec['dtype']}"
    extra = set(got) - {k for k, _ in leaves(want)}
    assert extra <= set(allowed_extra), f"keys the reference does not return: {extra}"


def test_manifest_is_the_reference_key_tree():
    """Shape of the fixture itself (CPU): both predictor settings, the three trees, the transform's keys."""
    for tp in (0, 1):
        m = MANIFEST[f"use_TP_net={tp}"]
        assert set(m) == {"specs", "reset", "step"}
        assert ("agents", "observation", "state_self") in dict(leaves(m["reset"]))
        assert ("next", "agents", "reward") in dict(leaves(m["step"])) and ("next", "done") in dict(leaves(m["step"]))
        assert ("truncated",) in dict(leaves(m["reset"]))
    assert dict(leaves(MANIFEST["use_TP_net=1"]["step"]))[("next", "agents", "observation", "state_self")]["shape"][-1] == 35


@pytest.mark.gpu
@pytest.mark.parametrize("use_tp", [0, 1])
def test_reset_and_step_return_the_reference_key_tree(use_tp):
    from hns_amd.env import HideAndSeek
    m = MANIFEST[f"use_TP_net={use_tp}"]
    A, C, E = MANIFEST["meta"]["num_agents"], MANIFEST["meta"]["num_cylinders"], 128
    cfg = config.make_cfg({"num_agents": A, "cylinder": {"max_num": C, "min_num": 4}, "env": {"num_envs": E, "max_episode_length": 30}},
                          algo={"use_TP_net": use_tp})
    env = HideAndSeek(cfg, headless=True)
    env.set_seed(0)
    td = env.reset()
    check_tree(m["reset"], td, E, allowed_extra={("done",)})                    # EnvBase.reset adds `done`
    inp = env.rand_step_input()
    out = env.step(inp)
    # what PIDRateController._inv_call leaves on the input tensordict (transforms.py:441,443) is there as well
    assert out[("stats", "action_error_order1")].shape == (E, A) and out[("info", "prev_action")].shape == (E, A, 4)
    check_tree({"next": m["step"]["next"]}, {"next": out["next"]} if not hasattr(out, "select") else out.select("next"), E)
    # spec trees: every entry of the reference's specs with the same shape / dtype
    ours = {"observation_spec": spec_tree(env.observation_spec), "action_spec": spec_tree(env.action_spec), "reward_spec": spec_tree(env.reward_spec)}
    for name in ("observation_spec", "action_spec", "reward_spec"):
        for key, rec in leaves(m["specs"][name]):
            node = ours[name]
            for k in key:
                assert k in node, f"{name}: missing {key}"
                node = node[k]
            shape = [E if (i == 0 and d == ME) else d for i, d in enumerate(rec["shape"])]
            assert node == [shape, rec["dtype"]], f"{name}{key}: {node} != {[shape, rec['dtype']]}"
    a = env.agent_spec["drone"]
    ref = m["specs"]["agent_spec"]
    assert (a.name, a.n) == (ref["name"], ref["n"])
    assert list(a.observation_key) == ref["observation_key"] and list(a.action_key) == ref["action_key"]
    assert list(a.reward_key) == ref["reward_key"] and list(a.state_key) == ref["state_key"]


@pytest.mark.gpu
def test_ctbr_and_target_rate_are_published_on_request():
    """transforms.py:456-457: with task.publish_ctbr the stepped tensordict also carries `ctbr` and `target_rate`; values
    bit-identical to the oracle's controller."""
    import numpy as np
    import hns_oracle as O
    from hns_amd.env import HideAndSeek
    cfg = config.make_cfg({"num_agents": 3, "publish_ctbr": 1, "cylinder": {"max_num": 5, "min_num": 4}, "env": {"num_envs": 128, "max_episode_length": 30}})
    env = HideAndSeek(cfg, headless=True)
    env.set_seed(2)
    env.reset()
    host = env.export_state()
    act = torch.randn(128, 3, 4, generator=torch.Generator().manual_seed(3))
    out = env.step(env.rand_step_input(act.to(env.device)))
    O.step(env.hcfg, host, act.numpy())
    assert out["ctbr"].shape == (128, 3, 4) and out["target_rate"].shape == (128, 3, 3)
    assert np.array_equal(out["ctbr"].cpu().numpy(), host["ctbr"]) and np.array_equal(out["target_rate"].cpu().numpy(), host["target_rate"][..., :3])
    assert np.array_equal(env.export_state()["drone_state"], host["drone_state"])


def test_task_yaml_for_the_unedited_train_script():
    """cfg/task/HideAndSeek_hip.yaml = the reference's task file with `action_transform: none` (train.py:160-176 then adds
    no controller transform) under a name the REGISTRY knows; it resolves to the same hns_cfg as the built-in defaults."""
    from hns_amd import abi
    root = os.path.dirname(HERE)
    cfg = config.load_cfg(os.path.join(root, "cfg", "task", "HideAndSeek_hip.yaml"))
    assert cfg.task.name == "HideAndSeek_hip" and str(cfg.task.action_transform).lower() == "none"
    ref_defaults = config.make_cfg({"env": {"num_envs": 65536}})
    a, b = config.resolve_hns_cfg(cfg), config.resolve_hns_cfg(ref_defaults)
    import ctypes as C
    assert bytes(C.string_at(C.addressof(a), C.sizeof(abi.HnsCfg))) == bytes(C.string_at(C.addressof(b), C.sizeof(abi.HnsCfg)))
    ref_yaml = "/root/reference/cfg/task/HideAndSeek.yaml"
    if os.path.exists(ref_yaml):                               # the authoring container only
        import yaml
        ref, ours = yaml.safe_load(open(ref_yaml)), yaml.safe_load(open(os.path.join(root, "cfg", "task", "HideAndSeek_hip.yaml")))
        assert ours["defaults"][0] == "HideAndSeek"            # hydra: inherits the reference's own task file
        # (besides the three overrides: the switches of this build, spelled out at their defaults — none of them a key of the reference's file)
        assert set(ours) - {"defaults"} == {"name", "action_transform", "env", "publish_ctbr", "pid_reset", "reset_extra_step"} and "action_transform" in ref
        assert not ({"publish_ctbr", "pid_reset", "reset_extra_step"} & set(ref)) and ours["pid_reset"] == "reference" and ours["reset_extra_step"] == 1
        # ... and what the file leaves out resolves, without hydra, to the reference file's values (the built-in defaults are those)
        full = config.load_cfg(ref_yaml, name="HideAndSeek_hip", action_transform="none", env={"num_envs": 65536, "max_episode_length": ref["env"]["max_episode_length"]})
        c = config.resolve_hns_cfg(full)
        assert bytes(C.string_at(C.addressof(a), C.sizeof(abi.HnsCfg))) == bytes(C.string_at(C.addressof(c), C.sizeof(abi.HnsCfg)))
