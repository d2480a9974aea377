"""Import shim: the package directory is named `rnb-neus-fork_amd` (not a valid Python identifier), so
`import rnb_neus_fork_amd` loads it from there and registers it under this importable name."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "rnb-neus-fork_amd")
_spec = importlib.util.spec_from_file_location(
    "rnb_neus_fork_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rnb_neus_fork_amd"] = _mod
_spec.loader.exec_module(_mod)
