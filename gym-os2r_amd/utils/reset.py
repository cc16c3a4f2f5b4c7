"""Closed-form leg IK used by the reset poses (mirror of gym_os2r/utils/reset.py:4-40)."""
import numpy as np

_REQUIRED = ["planarizer_pitch_joint", "upper_leg_length", "lower_leg_length",
             "central_pivot_height", "length_boom", "hip_offset", "clipping_adjust"]


def leg_joint_angles(robot_def: dict):
    """(hip, knee) angles [rad] that rest the foot on the ground for the given boom pitch.

    Lengths are in millimetres as in the settings tree; ``[0, 0]`` when the hip is too high
    for the leg to reach the ground (triangle inequality).
    """
    if not set(robot_def.keys()).issubset(set(_REQUIRED)):
        raise RuntimeError("One or more of the required params" + str(_REQUIRED)
                           + "were not provided for finding reset positions. ")
    lb, bp = robot_def["length_boom"], robot_def["planarizer_pitch_joint"]
    lh = (lb * np.sin(bp) + robot_def["central_pivot_height"]) / np.cos(bp)
    ul, ll = robot_def["upper_leg_length"], robot_def["lower_leg_length"]
    lleg = lh - robot_def["hip_offset"] - robot_def["clipping_adjust"]
    if lleg > ul + ll:
        return [0, 0]
    upper = np.arccos((ul ** 2 + lleg ** 2 - ll ** 2) / (2 * ul * lleg))
    lower = np.arcsin(ul * np.sin(upper) / ll) + upper
    return [upper, -lower]
