"""Settings tree with the reference's xpath accessors.

Mirrors ``gym_os2r.models.config.BaseConfig / SettingsConfig``
(gym_os2r/models/config/__init__.py:8-57): ``get_config('task_modes/fixed_hip/spaces')``
returns a shallow copy of the addressed node, ``set_config(value, xpath)`` creates
missing intermediate nodes.  The default tree is the compiled
``assets/settings.json`` (same keys and values as the reference's
``models/config/default/settings.yaml``); a YAML file can be loaded instead.
"""
from __future__ import annotations

import json
import os
from copy import copy
from functools import reduce
from operator import getitem

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


class BaseConfig:
    def __init__(self, path: str):
        if not os.path.isabs(path):
            path = os.path.join(_ASSETS, path)
        with open(path) as f:
            if path.endswith((".yaml", ".yml")):
                import yaml
                self.config_dict = yaml.load(f, Loader=yaml.FullLoader)
            else:
                self.config_dict = json.load(f)

    def set_config(self, value, xpath: str):
        keys = xpath.strip("/").split("/")
        d = self.config_dict
        for k in keys[:-1]:
            try:
                d = d[k]
            except KeyError:
                d[k] = {}
                d = d[k]
        d[keys[-1]] = value

    def get_config(self, xpath: str):
        keys = xpath.strip("/").split("/")
        return copy(reduce(getitem, keys[:-1], self.config_dict)[keys[-1]])


class SettingsConfig(BaseConfig):
    def __init__(self, path: str = "settings.json"):
        super().__init__(path)
