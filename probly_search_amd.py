"""Import shim: `import probly_search_amd` loads the package directory `probly-search_amd/`
(the hyphen of the reference's crate name is not a legal Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probly-search_amd")
_spec = importlib.util.spec_from_file_location("probly_search_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["probly_search_amd"] = _mod
_spec.loader.exec_module(_mod)
