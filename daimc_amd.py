"""Import shim: registers the package directory `deep-active-inference-mc_amd/` (not a valid Python
identifier) under the importable name `daimc_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'deep-active-inference-mc_amd')
_spec = importlib.util.spec_from_file_location('daimc_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['daimc_amd'] = _mod
_spec.loader.exec_module(_mod)
