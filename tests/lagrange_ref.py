"""Independent (non-spatial-algebra) mechanics of the compiled chain, for checking the oracle.

Route: forward kinematics -> centre-of-mass positions and body rotations -> geometric
Jacobians -> M(q) = sum_b m_b Jv^T Jv + Jw^T (R Ic R^T) Jw and V(q) = -sum m_b g.c_b.
Deliberately shares nothing with the ABA code path except the compiled model numbers.
"""
import numpy as np


def _rot(axis, q):
    c, s = np.cos(q), np.sin(q)
    if axis == 0:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == 1:
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def fk(model, q):
    """World rotation R[b] and origin o[b] of every body."""
    R, o = np.eye(3), np.zeros(3)
    Rs, os_ = [], []
    for i in range(model["nq"]):
        o = o + R @ np.array(model["rpos"][i])
        R = R @ np.array(model["rfix"][i]).reshape(3, 3) @ _rot(model["axis"][i], q[i])
        Rs.append(R.copy()); os_.append(o.copy())
    return Rs, os_


def _sym(ic):
    return np.array([[ic[0], ic[1], ic[2]], [ic[1], ic[3], ic[4]], [ic[2], ic[4], ic[5]]])


def mass_matrix(model, q, mass_scale=None):
    """M(q) from geometric Jacobians: column j of body b's COM Jacobian is a_j x (c_b - o_j),
    of its angular Jacobian a_j (world joint axis), for joints j on the path to b."""
    n = model["nq"]
    q = np.asarray(q, dtype=float)
    ms = np.ones(n) if mass_scale is None else np.asarray(mass_scale)
    R, o = fk(model, q)
    axes = [R[j][:, model["axis"][j]] for j in range(n)]
    M = np.zeros((n, n))
    for b in range(n):
        c = o[b] + R[b] @ np.array(model["com"][b])
        Jv = np.zeros((3, n)); Jw = np.zeros((3, n))
        for j in range(b + 1):
            Jv[:, j] = np.cross(axes[j], c - o[j])
            Jw[:, j] = axes[j]
        Iw = R[b] @ _sym(model["icom"][b]) @ R[b].T
        M += ms[b] * model["mass"][b] * Jv.T @ Jv + Jw.T @ Iw @ Jw
    return M


def potential(model, q, g=-9.8, mass_scale=None):
    ms = np.ones(model["nq"]) if mass_scale is None else np.asarray(mass_scale)
    R, o = fk(model, q)
    return sum(-ms[b] * model["mass"][b] * g * (o[b] + R[b] @ np.array(model["com"][b]))[2]
               for b in range(model["nq"]))


def grad_potential(model, q, g=-9.8, h=1e-6):
    q = np.asarray(q, dtype=float)
    out = np.zeros(len(q))
    for j in range(len(q)):
        dq = np.zeros(len(q)); dq[j] = h
        out[j] = (potential(model, q + dq, g) - potential(model, q - dq, g)) / (2 * h)
    return out


def kinetic(model, q, qd, mass_scale=None):
    M = mass_matrix(model, q, mass_scale)
    return 0.5 * np.asarray(qd) @ M @ np.asarray(qd)
