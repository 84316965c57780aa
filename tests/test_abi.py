"""CPU-side checks of the C-ABI boundary: the HIP library builds, loads without a GPU, exports
every symbol include/hns.h declares, and its struct layout matches the ctypes mirror."""
import ctypes
import os
import re

import pytest

from hns_amd import abi, config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    return ctypes.CDLL(g.build())


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "hns.h")).read()
    declared = set(re.findall(r"\b(hns_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(abi.EXPORTED_SYMBOLS), declared ^ set(abi.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_struct_layout_matches(lib):
    lib.hns_cfg_size.restype = ctypes.c_size_t
    assert lib.hns_cfg_size() == ctypes.sizeof(abi.HnsCfg)
    lib.hns_abi_version.restype = ctypes.c_int
    assert lib.hns_abi_version() == abi.HNS_ABI_VERSION
    assert ctypes.sizeof(abi.HnsBuffers) == 8 * len(abi.BUFFER_FIELDS)


def test_create_rejects_bad_config_without_gpu(lib):
    lib.hns_create.argtypes = [ctypes.POINTER(abi.HnsCfg), ctypes.POINTER(ctypes.c_void_p)]
    lib.hns_last_error.restype = ctypes.c_char_p
    c = config.resolve_hns_cfg(config.make_cfg())
    env = ctypes.c_void_p()
    bad = c.copy()
    bad.num_agents = 9
    assert lib.hns_create(ctypes.byref(bad), ctypes.byref(env)) == -1
    assert b"out of range" in lib.hns_last_error()
    bad = c.copy()
    bad.abi_version = 99
    assert lib.hns_create(ctypes.byref(bad), ctypes.byref(env)) == -1
    import torch
    if not torch.cuda.is_available():
        # valid config, but no device: the product path fails loudly instead of falling back to CPU
        assert lib.hns_create(ctypes.byref(c), ctypes.byref(env)) == -4
        assert b"no HIP device" in lib.hns_last_error()


def test_env_class_refuses_cpu():
    import torch
    from hns_amd.env import HideAndSeek, HnsError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(HnsError):
        HideAndSeek(config.make_cfg(), headless=True)


def test_config_schema_and_validation(tmp_path):
    # a task file written for the reference (same keys) loads unchanged
    y = tmp_path / "HideAndSeek.yaml"
    y.write_text("name: HideAndSeek\ndefaults:\n  - /base/env_base@_here_\nenv:\n  num_envs: 128\n  max_episode_length: 100\n"
                 "num_agents: 4\narena_size: 0.9\ncylinder:\n  size: 0.1\n  fixed_num: 2\n  min_num: 0\n  max_num: 6\n  obs_max_cylinder: 3\n")
    cfg = config.load_cfg(str(y))
    c = config.resolve_hns_cfg(cfg)
    assert (c.num_envs, c.num_agents, c.num_cylinders, c.cyl_fixed_num, c.max_episode_length) == (128, 4, 6, 2, 100)
    assert abs(c.v_prey - 1.3) < 1e-6 and c.grid_num == 9
    config.resolve_hns_cfg(config.make_cfg(algo={"use_TP_net": 1}))          # TP_net runs above the kernel (tp_net.py)
    assert config.resolve_hns_cfg(config.make_cfg({"use_obstacles": 1}, algo={"use_TP_net": 1})).tp_use_obstacles == 1   # 7+9+15 = 31 values
    assert config.resolve_hns_cfg(config.make_cfg({"use_obstacles": 1})).tp_use_obstacles == 0                           # only read with TP_net
    assert config.resolve_hns_cfg(config.make_cfg({"use_obstacles": 1, "cylinder": {"max_num": 8}}, algo={"use_TP_net": 1})).tp_use_obstacles == 1   # 40 values: three chunks
    wide = config.resolve_hns_cfg(config.make_cfg({"num_agents": 7, "use_obstacles": 1, "cylinder": {"max_num": 16}}, algo={"use_TP_net": 1}))
    assert wide.tp_use_obstacles == 1 and abi.tp_frame_dim(7, 16, 1) == 76  # the widest frame the shapes allow: five 16-value chunks
    with pytest.raises(ValueError):
        config.resolve_hns_cfg(config.make_cfg({"cylinder": {"max_num": 40}}))
    sc = config.resolve_hns_cfg(config.make_cfg({"use_random_cylinder": 0, "scenario_flag": "wall"}))
    assert sc.init_mode == abi.HNS_INIT_SCENARIO and sc.fixed_cyl_active == 4


@pytest.mark.gpu
def test_clock_probe_reads_a_plausible_shader_clock():
    """hns_clock_probe (bench.py's clock_mhz_* fields): shader cycles per 100 MHz tick of one spinning wave — between 0.5 and 2.6 GHz on an MI355X (maximum 2.4 GHz + boost
    margin), two probes of an idle chip within a few percent of each other; bad arguments are refused."""
    import torch
    from hns_amd.env import HideAndSeek
    env = HideAndSeek(config.make_cfg({"env": {"num_envs": 64}}), headless=True)
    a, b = env.clock_probe(2000), env.clock_probe(2000)
    torch.cuda.synchronize()
    ma, mb = env.clock_mhz(a), env.clock_mhz(b)
    assert 500.0 < ma < 2600.0 and 500.0 < mb < 2600.0 and abs(ma - mb) / ma < 0.1, (ma, mb)
    ticks = int(a[1])
    assert 2000 <= ticks < 4000                                   # it span for the 20 us asked for
    assert env._lib.hns_clock_probe(None, 2000, None) != 0 and env._lib.hns_clock_probe(a.data_ptr(), 0, None) != 0
