"""Import alias: `import gsmcal` loads the package directory multi-rtl-sdr-calibration_amd/
(its mandated name contains hyphens, so it cannot be imported by name)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_d = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "multi-rtl-sdr-calibration_amd")
_spec = _ilu.spec_from_file_location("gsmcal", _os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["gsmcal"] = _mod
_spec.loader.exec_module(_mod)
