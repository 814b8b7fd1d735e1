"""Import alias: `import mdm_amd` loads the package that lives in `motion-diffusion-model_amd/`
(the directory name the project layout prescribes is not a valid Python identifier)."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "motion-diffusion-model_amd")
_spec = importlib.util.spec_from_file_location(
    "mdm_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mdm_amd"] = _mod
_spec.loader.exec_module(_mod)
