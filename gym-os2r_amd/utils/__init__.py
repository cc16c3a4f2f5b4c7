from . import reset

__all__ = ["reset"]
