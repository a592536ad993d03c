"""Import alias: the package directory `multi-robot-fabrics_amd/` has a hyphen in its name, which the
import system cannot spell; this module loads it under the importable name `multi_robot_fabrics_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi-robot-fabrics_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
