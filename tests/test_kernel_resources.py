"""Every kernel the library ships: no spilled vector register, no scratch memory.

The metadata of the gfx950 code objects inside build/obj/*.o is read as tools/kernel_resources.py reads it (llvm-objcopy -> clang-offload-bundler ->
llvm-readelf --notes): `.vgpr_spill_count` and `.private_segment_fixed_size` of every kernel must be 0.  Round 5 shipped 11 step kernels that spilled (two-evader
instantiations forced under 128 VGPRs) and reset kernels with 44 bytes of indexed scratch (the Philox state); DESIGN's "0 spilled registers" was true of the
predictor table only.  CPU test: hipcc cross-compiles here, no GPU needed."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# kernels allowed to use scratch, with the reason: empty — keep it that way or say why here
ALLOWED = {}


@pytest.fixture(scope="module")
def kernels():
    import kernel_resources
    objdir = os.path.join(ROOT, "build", "obj")
    lib = os.path.join(ROOT, "multi-uav-pursuit-evasion_amd", "libhns.so")
    stale = not os.path.isdir(objdir) or not os.listdir(objdir) or not os.path.exists(lib) or \
        max(os.path.getmtime(os.path.join(objdir, f)) for f in os.listdir(objdir)) < os.path.getmtime(lib) - 600
    if stale:                                                   # objects missing or older than the library they should have made: rebuild both
        import __graft_entry__
        __graft_entry__.build(force=True)
    return kernel_resources.all_kernels(objdir)


def test_no_kernel_spills_or_uses_scratch(kernels):
    assert len(kernels) > 150                                   # 7 pursuer counts x (tile + small + reset instantiations) + predictor + generator + helpers
    names = {k["demangled"] for k in kernels}
    for must in ("hns_step_v4_kernel<3, 1, false, 4, false, 8, false>", "hns_step_v4_kernel<6, 2, false, 4, false, 16, false>",
                 "hns_step_small_kernel<3, false, 0>", "hns_tp_lstm_ws_kernel<1, 4>", "hns_tp_lstm_ws_kernel<1, 2>", "hns_tp_lstm_ws_kernel<1, 1>",
                 "hns_tp_lstm_ws_kernel<2, 1>", "hns_tp_lstm_ws_kernel<5, 1>", "hns_tp_lstm_ws_kernel<5, 4>",
                 "hns_reset_kernel<3, 1, 4>",
                 "hns_step_v4_kernel<3, 1, true, 4, false, 0, true>"):
        assert must in names, f"{must} is not in the library's objects"
    bad = [(k["demangled"], k["vgpr_spill_count"], k["private_segment_fixed_size"]) for k in kernels
           if (k["vgpr_spill_count"] or k["private_segment_fixed_size"]) and k["demangled"] not in ALLOWED]
    assert not bad, "kernels with spilled vector registers / scratch bytes: " + "; ".join(f"{n}: {v} VGPRs spilled, {s} B scratch" for n, v, s in bad)


def test_register_budgets_of_the_headline_kernels(kernels):
    """Occupancy the design counts on: the 3v1 tile kernel runs four 4-wave workgroups per CU (<= 128 VGPRs), the 6v2 shard two 7-wave
    workgroups (<= 128), the small-batch mapping two 7-wave workgroups (<= 128), the predictor — every tile count of the one-chunk frame — four waves per SIMD
    (<= 128: two 8-wave workgroups per CU)."""
    by = {k["demangled"]: k for k in kernels}
    assert by["hns_step_v4_kernel<3, 1, false, 4, false, 8, false>"]["vgpr_count"] <= 96
    assert by["hns_step_v4_kernel<6, 2, false, 4, false, 16, false>"]["vgpr_count"] <= 128
    assert by["hns_step_small_kernel<3, false, 8>"]["vgpr_count"] <= 128
    for nxc, tiles in ((1, 1), (1, 2), (1, 4), (2, 1)):       # (two-chunk frames: the one-tile kernel; the four-tile one runs two waves per SIMD)
        tp = by[f"hns_tp_lstm_ws_kernel<{nxc}, {tiles}>"]
        assert tp["vgpr_count"] + tp["agpr_count"] <= 128
