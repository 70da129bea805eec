"""Import shim: the package directory is ``piccolo.jl_amd/`` (a dot is not importable as a
module name), so this loads it under the name ``piccolo_jl_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "piccolo.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "piccolo_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["piccolo_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
