from .. import config  # noqa: F401  (gym_os2r.models.config)
from . import monopod

__all__ = ["monopod", "config"]
