"""Import alias: ``import idealnerf_amd`` loads the package in ``ideal-nerf_amd/``
(a directory name Python cannot import directly)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ideal-nerf_amd")
_spec = importlib.util.spec_from_file_location("idealnerf_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["idealnerf_amd"] = _mod
_spec.loader.exec_module(_mod)
