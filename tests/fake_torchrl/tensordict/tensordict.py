from . import TensorDict, TensorDictBase  # noqa: F401  (omni_drones/utils/torchrl/collector.py:29 imports from here)
