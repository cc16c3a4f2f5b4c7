"""Closed-form leg inverse kinematics behind the reset poses (what gym_os2r/utils/reset.py:4-40 computes)."""
import numpy as np

_KNOWN_KEYS = ("planarizer_pitch_joint", "upper_leg_length", "lower_leg_length",
               "central_pivot_height", "length_boom", "hip_offset", "clipping_adjust")


def _fold(thigh, shank, reach):
    """Interior angles of the thigh / shank / hip-to-foot triangle: law of cosines at the hip, then law of
    sines for the knee (same operation order as the reference, so the values agree bit for bit)."""
    at_hip = np.arccos((thigh ** 2 + reach ** 2 - shank ** 2) / (2 * thigh * reach))
    at_knee = np.arcsin(thigh * np.sin(at_hip) / shank) + at_hip
    return at_hip, at_knee


def leg_joint_angles(robot_def: dict):
    """(hip, knee) joint angles [rad] that put the foot on the ground for the boom pitch in ``robot_def``.

    Lengths are millimetres, as in the settings tree.  A leg too short to reach the ground from the hip
    height (triangle inequality) yields ``[0, 0]``; a key outside the known set raises ``RuntimeError``.
    """
    unknown = set(robot_def) - set(_KNOWN_KEYS)
    if unknown:
        raise RuntimeError(f"reset pose definition has keys outside {list(_KNOWN_KEYS)}: {sorted(unknown)}")
    pitch = robot_def["planarizer_pitch_joint"]
    hip_height = (robot_def["length_boom"] * np.sin(pitch) + robot_def["central_pivot_height"]) / np.cos(pitch)
    thigh, shank = robot_def["upper_leg_length"], robot_def["lower_leg_length"]
    reach = hip_height - robot_def["hip_offset"] - robot_def["clipping_adjust"]
    if reach > thigh + shank:
        return [0, 0]
    at_hip, at_knee = _fold(thigh, shank, reach)
    return [at_hip, -at_knee]
