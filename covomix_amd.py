"""Import shim: `import covomix_amd` loads the package that lives in the
directory `neurips2024-covomix_amd/` (a name Python cannot import directly)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "neurips2024-covomix_amd")
_spec = importlib.util.spec_from_file_location(
    "covomix_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["covomix_amd"] = _mod
_spec.loader.exec_module(_mod)
