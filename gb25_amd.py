"""Import shim: the package directory is `gb-25_amd/` (not a valid identifier), so
`import gb25_amd` loads it from there under the importable name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gb-25_amd")
_spec = importlib.util.spec_from_file_location("gb25_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gb25_amd"] = _mod
_spec.loader.exec_module(_mod)
