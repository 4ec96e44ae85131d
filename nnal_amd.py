"""Import shim: registers the package directory `nn-active-learning_amd/` (not a valid Python
identifier) as the module `nnal_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nn-active-learning_amd')
_spec = importlib.util.spec_from_file_location('nnal_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['nnal_amd'] = _mod
_spec.loader.exec_module(_mod)
