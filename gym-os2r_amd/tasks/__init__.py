from . import monopod, monopod_no_norm

__all__ = ["monopod", "monopod_no_norm"]
