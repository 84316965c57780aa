"""Bootstrap for `import omni_drones` on a box WITHOUT Isaac Sim ("Running the reference's scripts/train.py unedited").

The reference's package initialiser imports `omni.isaac.kit.SimulationApp` (omni_drones/__init__.py:27) and `scripts/train.py` calls
`init_simulation_app(cfg)` at :96 and `simulation_app.close()` at :323.  This package is put on `sys.path` BEFORE the reference checkout:

* it provides what `scripts/train*.py` import from the top level — `CONFIG_PATH` and `init_simulation_app` (an object with `.close()`;
  there is no simulator to launch, the HIP env owns the physics) — and the `TensorDict.shapes` / `.devices` properties the reference
  attaches there (:49-63, read by `IsaacEnv.__init__`'s pprint and by user scripts);
* `omni_drones.envs` / `omni_drones.envs.isaac_env` (this directory) give `IsaacEnv.REGISTRY` (isaac_env.py:52,154-161 — the reference's own
  plug-in point, looked up at scripts/train.py:110-111) holding the HIP classes, instead of the reference's `envs/__init__.py`, which imports Isaac;
* every OTHER submodule (`omni_drones.utils.torchrl`, `omni_drones.learning`, `omni_drones.controllers`, ...) is found in the reference
  checkout named by `OMNI_DRONES_SRC` (the directory that holds the reference's `omni_drones/`), appended to this package's `__path__`.

Nothing here is copied from the reference; it is a written-from-scratch shim of five names."""
import os

_src = os.environ.get("OMNI_DRONES_SRC")
if _src:
    _pkg = os.path.join(_src, "omni_drones")
    if not os.path.isdir(_pkg):
        raise ImportError(f"OMNI_DRONES_SRC={_src!r} holds no omni_drones/ directory")
    __path__.append(_pkg)                                   # this directory first (envs/ is ours), the reference's tree behind it
    CONFIG_PATH = os.path.join(_src, "cfg")
else:
    CONFIG_PATH = os.environ.get("OMNI_DRONES_CFG", os.path.join(os.getcwd(), "cfg"))


class _NoSimulationApp:
    """What `init_simulation_app` returns here: scripts/train.py only keeps it to call `.close()` at the end (:96, :323)."""

    def __init__(self, cfg):
        self.config = {"headless": bool(cfg["headless"]) if "headless" in cfg else True}

    def is_running(self):
        return True

    def update(self):
        pass

    def close(self):
        pass


def init_simulation_app(cfg):
    return _NoSimulationApp(cfg)


try:                                                        # omni_drones/__init__.py:49-63 — `.shapes` / `.devices` on TensorDict
    import torch as _torch
    from tensordict import TensorDict as _TD

    if not hasattr(_TD, "shapes"):
        _TD.shapes = property(lambda self: {k: v.shape if isinstance(v, _torch.Tensor) else v.shapes for k, v in self.items()})
    if not hasattr(_TD, "devices"):
        _TD.devices = property(lambda self: {k: v.device if isinstance(v, _torch.Tensor) else v.devices for k, v in self.items()})
except ImportError:                                         # no tensordict on this box: hns_amd falls back to its own record type
    pass
