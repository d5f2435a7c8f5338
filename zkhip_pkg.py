"""Imports the package directory `zkevm-circuits_amd/` (not a valid identifier) as `zkevm_circuits_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    name = "zkevm_circuits_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg_dir = os.path.join(_ROOT, "zkevm-circuits_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
