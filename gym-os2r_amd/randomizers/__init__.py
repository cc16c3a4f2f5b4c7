from . import monopod, monopod_no_rand

__all__ = ["monopod", "monopod_no_rand"]
