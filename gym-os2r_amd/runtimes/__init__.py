from . import hip_runtime
from .hip_runtime import HipRuntime

__all__ = ["hip_runtime", "HipRuntime"]
