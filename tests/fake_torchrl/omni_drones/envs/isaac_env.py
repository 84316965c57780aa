"""`from omni_drones.envs.isaac_env import IsaacEnv` (scripts/train.py:100) → `IsaacEnv.REGISTRY[cfg.task.name]` (:110-111).

The registry is the reference's plug-in point (isaac_env.py:52,154-161); here it holds the HIP classes under the reference's task names
and under `HideAndSeek_hip` (cfg/task/HideAndSeek_hip.yaml)."""
import hns_amd  # noqa: F401  (alias for the package directory, whose name is not an identifier)
from hns_amd.env import HideAndSeek
from hns_amd.envgen import HideAndSeek_envgen
from hns_amd.hover import Hover


class IsaacEnv:
    REGISTRY = {}


for _name, _cls in (("HideAndSeek", HideAndSeek), ("HideAndSeek_hip", HideAndSeek), ("HideAndSeek_envgen", HideAndSeek_envgen),
                    ("HideAndSeek_envgen_hip", HideAndSeek_envgen), ("Hover", Hover)):
    IsaacEnv.REGISTRY[_name] = _cls
    IsaacEnv.REGISTRY[_name.lower()] = _cls
