"""MI355X-native batched monopod stepper behind the gym-os2r task / runtime API.

The package directory is ``gym-os2r_amd/`` (the repository layout contract); it is
imported as ``gym_os2r_amd`` through the alias module at the repository root.
Sub-modules that need the HIP extension load it on first use and raise if it is
missing -- there is no CPU fallback.
"""
import json as _json
import os as _os

from . import abi, config, rewards, spaces, tasks, utils  # noqa: F401

__version__ = "0.1.0"

_ASSETS = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "assets")


def load_models() -> dict:
    """Compiled chain models keyed by the reference's model names ('monopod', ...)."""
    with open(_os.path.join(_ASSETS, "models.json")) as f:
        return _json.load(f)["models"]


def get_model(name: str) -> dict:
    models = load_models()
    if name not in models:
        raise KeyError(f"unknown monopod model {name!r}; available: {sorted(models)}")
    return models[name]


from . import common, models, randomizers, registry, runtimes, scenario  # noqa: F401,E402
from .registry import REGISTRY, make  # noqa: F401,E402
