"""Import alias for the package directory `multi-uav-pursuit-evasion_amd/`.

The directory name is fixed by the project layout and is not a valid Python identifier, so
it is loaded here under the canonical module name ``hns_amd`` (``import hns_amd``,
``from hns_amd.env import HideAndSeek``).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi-uav-pursuit-evasion_amd")
_spec = importlib.util.spec_from_file_location(
    "hns_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hns_amd"] = _mod
_spec.loader.exec_module(_mod)
