"""The branch of hns_amd that subclasses torchrl's EnvBase (taken when `torchrl` / `tensordict` import) — executed against
tests/fake_torchrl, stand-ins that enforce what torchrl 0.1.1 / tensordict 0.1.2 enforce (the real packages cannot be installed
in the build image).  Reference wiring: omni_drones/envs/isaac_env.py:47-57,210-240, scripts/train.py:165-205,
omni_drones/utils/torchrl/collector.py:33-87."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_torchrl")


def test_stand_ins_enforce_the_contract():
    """CPU: the checks themselves — a spec without the batch shape, a missing key without default, a new key on a locked tensordict."""
    code = r"""
import sys; sys.path.insert(0, %r)
import torch
from tensordict import TensorDict
from torchrl.data import CompositeSpec, UnboundedContinuousTensorSpec
from torchrl.envs import EnvBase
class E(EnvBase):
    def _set_seed(self, s): pass
e = E(device="cpu", batch_size=[4])
try:
    e.observation_spec = CompositeSpec({"o": UnboundedContinuousTensorSpec((4, 3))})       # composite shape () != batch (4,)
    raise SystemExit("spec without the batch shape was accepted")
except ValueError:
    pass
e.observation_spec = CompositeSpec({"o": UnboundedContinuousTensorSpec((4, 3))}, shape=[4])
td = TensorDict({"a": torch.zeros(4, 2)}, [4])
try:
    td.get("_reset"); raise SystemExit("missing key returned something")
except KeyError:
    pass
assert td.get("_reset", None) is None
td.lock_()
try:
    td.set("b", torch.zeros(4)); raise SystemExit("locked tensordict took a new key")
except RuntimeError:
    pass
try:
    TensorDict({"a": torch.zeros(3, 2)}, [4]); raise SystemExit("wrong batch size accepted")
except RuntimeError:
    pass
print("ok")
""" % FAKE
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_omni_drones_bootstrap_without_isaac():
    """CPU: `import omni_drones` on a box without Isaac Sim (VERDICT r4 missing #2).  scripts/train.py:13 imports CONFIG_PATH and
    init_simulation_app from a package whose initialiser imports omni.isaac.kit (omni_drones/__init__.py:27), calls the latter at :96, looks the
    env class up in `IsaacEnv.REGISTRY[cfg.task.name]` (:100,:110-111) and closes the app at :323.  tests/fake_torchrl/omni_drones is the shim
    INTEGRATION.md tells a maintainer to put first on sys.path; with OMNI_DRONES_SRC naming a reference checkout every other submodule is found
    there (exercised only where /root/reference exists: it never travels to the GPU box)."""
    ref = "/root/reference"
    code = r"""
import os, sys
sys.path[:0] = [%r, %r]
import omni_drones
from omni_drones import CONFIG_PATH, init_simulation_app
app = init_simulation_app({"headless": True})
from omni_drones.envs.isaac_env import IsaacEnv
import hns_amd.env, hns_amd.envgen
assert IsaacEnv.REGISTRY["HideAndSeek_hip"] is hns_amd.env.HideAndSeek and IsaacEnv.REGISTRY["HideAndSeek"] is hns_amd.env.HideAndSeek
assert IsaacEnv.REGISTRY["HideAndSeek_envgen"] is hns_amd.envgen.HideAndSeek_envgen and "hover" in IsaacEnv.REGISTRY
from tensordict import TensorDict
import torch
td = TensorDict({"a": torch.zeros(4, 2), "b": {"c": torch.zeros(4, 3)}}, [4])
assert td.shapes == {"a": torch.Size([4, 2]), "b": {"c": torch.Size([4, 3])}}, td.shapes
if os.environ.get("OMNI_DRONES_SRC"):
    assert os.path.samefile(CONFIG_PATH, os.path.join(os.environ["OMNI_DRONES_SRC"], "cfg")) and os.path.isfile(os.path.join(CONFIG_PATH, "train.yaml"))
    from omni_drones.utils.torch import quat_rotate                      # the reference's own module, found behind the shim
    from omni_drones.actuators.rotor_group import RotorGroup             # (omni_drones.learning needs the REAL tensordict: tensordict.utils)
    v = quat_rotate(torch.tensor([[1.0, 0.0, 0.0, 0.0]]), torch.tensor([[1.0, 2.0, 3.0]]))
    assert torch.equal(v, torch.tensor([[1.0, 2.0, 3.0]]))
    assert omni_drones.envs.isaac_env.__file__.startswith(%r)            # ... while omni_drones.envs stays the shim's
app.close()
print("ok")
""" % (FAKE, ROOT, FAKE)
    for src in ([None, ref] if os.path.isdir(os.path.join(ref, "omni_drones")) else [None]):
        env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "OMNI_DRONES_SRC")}
        if src:
            env["OMNI_DRONES_SRC"] = src
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("use_tp", [0, 1])
def test_env_under_transformed_env_and_collector(use_tp):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    out = subprocess.run([sys.executable, os.path.join(FAKE, "run_collector.py"), str(use_tp)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["rollouts"] == 4 and r["masked_resets"] >= 2 and r["use_tp"] == use_tp
    assert r["stats_keys"] == 24 and r["episodes_seen"] >= 2 * 256            # scripts/train.py:53-80,113-116: EpisodeStats' picks ≡ the oracle's finished-episode statistics
    assert (r["tp_updates"] >= 1) == bool(use_tp)                              # learning/mappo.py:407-427,252-268: the predictor trained on windows of the collected ground truth


@pytest.mark.gpu
def test_env_under_rollout_evaluate_and_a_learner():
    """scripts/train.py's other two consumers: `evaluate()`'s `env.rollout(..., auto_reset=True, break_when_any_done=False,
    return_contiguous=False)` in eval mode with a callback, and two PPO-style iterations of an attention policy that reads the observation in
    spec-key order; every trajectory replayed on the oracle bit for bit, reset_pid pulsing past the end of the episode."""
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    out = subprocess.run([sys.executable, os.path.join(FAKE, "run_rollout.py")], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["evaluate_steps"] == 13 and r["reset_pid_pulses"] == 3 * 192 and r["frames"] == 6 and r["ppo_iterations"] == 2
    assert r["video_shape"] == [6, 3, 128, 128] and r["eval_stats"] == 24                # scripts/train.py:236-254 ran to the video array
