"""Import alias: the package directory is named ``self-supervised-mvs_amd`` (not a valid Python
identifier), so ``import mvs_amd`` loads it under this name."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "self-supervised-mvs_amd")
_spec = importlib.util.spec_from_file_location("mvs_amd", os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mvs_amd"] = _mod
_spec.loader.exec_module(_mod)
