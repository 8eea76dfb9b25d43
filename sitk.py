"""Import shim: `import sitk` loads the package that lives in ./surface-vision-transformers_amd/
(a directory name that is not a Python identifier) under the module name `sitk`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "surface-vision-transformers_amd")
_spec = importlib.util.spec_from_file_location("sitk", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sitk"] = _mod
_spec.loader.exec_module(_mod)
