"""Import shim: the package directory is `add-gym_amd/` (repo naming); a hyphen cannot be
imported, so `import add_gym_amd` loads that directory under a valid module name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "add-gym_amd")
_spec = importlib.util.spec_from_file_location("add_gym_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["add_gym_amd"] = _mod
_spec.loader.exec_module(_mod)
