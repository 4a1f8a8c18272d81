"""Import shim: the package directory is named `cp-cals_amd` (not a Python identifier); this module
loads it under the importable name `cp_cals_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cp-cals_amd")
_spec = importlib.util.spec_from_file_location(
    "cp_cals_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cp_cals_amd"] = _mod
_spec.loader.exec_module(_mod)
