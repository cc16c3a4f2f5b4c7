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

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def _split(xpath: str):
    return [part for part in xpath.strip("/").split("/")]


class BaseConfig:
    """A nested dict addressed by '/'-separated paths (no wildcards, no type enforcement)."""

    def __init__(self, path: str):
        if not os.path.isabs(path):
            path = os.path.join(_ASSETS, path)
        with open(path) as f:
            if path.endswith((".yaml", ".yml")):
                import yaml
                self.config_dict = yaml.load(f, Loader=yaml.FullLoader)
            else:
                self.config_dict = json.load(f)

    def _parent(self, parts, create: bool):
        node = self.config_dict
        for name in parts[:-1]:
            node = node.setdefault(name, {}) if create else node[name]     # KeyError on a missing branch when reading
        return node

    def set_config(self, value, xpath: str):
        """Write ``value`` at ``xpath``; branches that do not exist yet are created."""
        parts = _split(xpath)
        self._parent(parts, create=True)[parts[-1]] = value

    def get_config(self, xpath: str):
        """Shallow copy of the node at ``xpath`` (``KeyError`` if it does not exist)."""
        parts = _split(xpath)
        return copy(self._parent(parts, create=False)[parts[-1]])


class SettingsConfig(BaseConfig):
    def __init__(self, path: str = "settings.json"):
        super().__init__(path)
