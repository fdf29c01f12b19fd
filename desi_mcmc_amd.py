"""Import alias: the package lives in `desi-mcmc_amd/` (a hyphen is not a valid identifier).

`import desi_mcmc_amd` executes this file, which loads the real package from that directory
under this name and replaces itself in sys.modules.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "desi-mcmc_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
