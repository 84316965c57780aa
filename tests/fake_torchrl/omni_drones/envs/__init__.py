"""`omni_drones.envs` without Isaac: the registry and the HIP task classes (the reference's envs/__init__.py imports its Isaac-backed tasks)."""
from .isaac_env import IsaacEnv, HideAndSeek, HideAndSeek_envgen, Hover  # noqa: F401
