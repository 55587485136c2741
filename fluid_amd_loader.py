"""Import the hyphen-named package directory ``fluid-simulation_amd/`` as the module
``fluid_simulation_amd`` (a hyphen is not a valid identifier, so plain ``import`` cannot)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "fluid-simulation_amd")
MODNAME = "fluid_simulation_amd"


def load():
    if MODNAME in sys.modules:
        return sys.modules[MODNAME]
    spec = importlib.util.spec_from_file_location(
        MODNAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[MODNAME] = mod
    spec.loader.exec_module(mod)
    return mod
