"""Import helper: the package directory is named `root-simple-mcmc_amd` (a hyphen is
not a valid Python identifier), so it is loaded by path and registered as
`root_simple_mcmc_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "root-simple-mcmc_amd")
_NAME = "root_simple_mcmc_amd"


def load_package():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
