"""Monopod task without observation normalisation.

Mirror of ``gym_os2r.tasks.monopod_no_norm.MonopodTask`` (tasks/monopod_no_norm.py:105-246):
observations are raw joint positions (periodic ones wrapped to [-pi, pi)) and raw
velocities, and the reward class is built with ``normalized=False``.
"""
from .monopod import MonopodTask as _NormalizedTask


class MonopodTask(_NormalizedTask):
    normalized = False
