"""Import alias for the package directory ``gym-os2r_amd/``.

The repository layout names the package after the reference (``gym-os2r`` + ``_amd``); a hyphen
is not a valid Python identifier, so this module loads that directory under the importable
name ``gym_os2r_amd`` and replaces itself in ``sys.modules`` with the real package.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gym-os2r_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
