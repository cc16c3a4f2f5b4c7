"""Model wrapper with the reference's constructor (gym_os2r/models/monopod.py:9-38): inserts one
of the compiled monopod variants into a (facade) world under a unique name."""
from typing import List

from .. import load_models
from ..scenario import Pose

_counter = {}


def get_unique_model_name(world, model_name: str) -> str:
    n = _counter.get(model_name, 0)
    name = model_name
    while name in world.model_names():
        n += 1
        name = f"{model_name}{n}"
    _counter[model_name] = n
    return name


def get_model_file_from_name(robot_name: str) -> str:
    if robot_name not in load_models():
        raise RuntimeError(f"Failed to find robot '{robot_name}'")
    return robot_name                                   # compiled assets are addressed by name


class Monopod:
    def __init__(self, world, monopod_version: str, position: List[float] = (0.0, 0.0, 0.0),
                 orientation: List[float] = (1.0, 0, 0, 0), model_file: str = None):
        model_name = get_unique_model_name(world, "monopod")
        if model_file is None:
            model_file = get_model_file_from_name(monopod_version)
        if not world.to_gazebo().insert_model(model_file, Pose(position, orientation), model_name):
            raise RuntimeError("Failed to insert model")
        self.model = world.get_model(model_name)

    def __getattr__(self, name):
        return getattr(self.model, name)
