"""URDF (+ binary STL collision meshes) -> flat serial-chain model for the HIP stepper.

The reference hands its URDFs (gym_os2r/models/models/<variant>/<variant>.urdf) to
sdformat/DART at run time (gym_os2r/models/monopod.py:27,
gym_os2r/randomizers/monopod.py:171).  Here the same description is compiled
once, on the host, into the numbers the kernels consume (``Os2rModel`` in
include/os2r.h):

* URDF joint transform  = Trans(xyz) . Rz(yaw) . Ry(pitch) . Rx(roll) . Rot(axis, q)
  with the literal rpy values of the file (1.57 is *not* pi/2 and is kept as is);
* ``type="fixed"`` joints are lumped into the parent body (composite inertia,
  collision points carried over), which is what sdformat's URDF->SDF conversion does;
  a body welded to the world is static and contributes nothing;
* each body's collision mesh is reduced to convex-hull vertices, pruned of
  points that can never reach the ground plane z=0, and decimated by
  farthest-point sampling to ``max_cand_per_link`` contact candidate points.

Only numpy is needed at run time for precompiled assets; scipy is imported
lazily when a mesh has to be hulled.
"""
from __future__ import annotations

import math
import os
import struct
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional, Sequence

import numpy as np

MAX_DOF = 5
MAX_CAND = 192


# ----------------------------------------------------------------------------
# small rotation helpers
# ----------------------------------------------------------------------------
def rpy_to_matrix(roll: float, pitch: float, yaw: float) -> np.ndarray:
    """URDF fixed-axis roll-pitch-yaw -> rotation matrix Rz(yaw) Ry(pitch) Rx(roll)."""
    cr, sr = math.cos(roll), math.sin(roll)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]], dtype=np.float64)
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]], dtype=np.float64)
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]], dtype=np.float64)
    return rz @ ry @ rx


def _floats(text: Optional[str], n: int, default: float = 0.0) -> np.ndarray:
    if text is None:
        return np.full(n, default, dtype=np.float64)
    vals = [float(t) for t in text.split()]
    if len(vals) != n:
        raise ValueError(f"expected {n} numbers, got {text!r}")
    return np.array(vals, dtype=np.float64)


# ----------------------------------------------------------------------------
# URDF parsing
# ----------------------------------------------------------------------------
def parse_urdf(path: str) -> dict:
    """Parse the subset of URDF the monopod descriptions use."""
    root = ET.parse(path).getroot()
    if root.tag != "robot":
        raise ValueError(f"{path}: not a URDF (<robot> root expected)")
    links: Dict[str, dict] = {}
    for le in root.findall("link"):
        link = {"name": le.get("name"), "mass": 0.0, "com": np.zeros(3),
                "inertia": np.zeros((3, 3)), "collisions": []}
        ine = le.find("inertial")
        if ine is not None:
            org = ine.find("origin")
            xyz = _floats(org.get("xyz") if org is not None else None, 3)
            rpy = _floats(org.get("rpy") if org is not None else None, 3)
            link["mass"] = float(ine.find("mass").get("value"))
            it = ine.find("inertia")
            ixx, ixy, ixz, iyy, iyz, izz = (float(it.get(k)) for k in
                                            ("ixx", "ixy", "ixz", "iyy", "iyz", "izz"))
            imat = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
            rin = rpy_to_matrix(*rpy)
            link["com"] = xyz
            link["inertia"] = rin @ imat @ rin.T  # expressed in link axes, about the COM
        for ce in le.findall("collision"):
            org = ce.find("origin")
            xyz = _floats(org.get("xyz") if org is not None else None, 3)
            rpy = _floats(org.get("rpy") if org is not None else None, 3)
            mesh = ce.find("geometry/mesh")
            if mesh is None:
                continue  # only mesh collisions appear in the monopod models
            link["collisions"].append({"xyz": xyz, "rot": rpy_to_matrix(*rpy),
                                       "mesh": mesh.get("filename")})
        links[link["name"]] = link
    joints: List[dict] = []
    for je in root.findall("joint"):
        org = je.find("origin")
        ax = je.find("axis")
        dyn = je.find("dynamics")
        joints.append({
            "name": je.get("name"),
            "type": je.get("type"),
            "parent": je.find("parent").get("link"),
            "child": je.find("child").get("link"),
            "xyz": _floats(org.get("xyz") if org is not None else None, 3),
            "rot": rpy_to_matrix(*_floats(org.get("rpy") if org is not None else None, 3)),
            "axis": _floats(ax.get("xyz") if ax is not None else "1 0 0", 3),
            "damping": float(dyn.get("damping", 0.0)) if dyn is not None else 0.0,
            "friction": float(dyn.get("friction", 0.0)) if dyn is not None else 0.0,
        })
    return {"name": root.get("name"), "links": links, "joints": joints}


# ----------------------------------------------------------------------------
# meshes
# ----------------------------------------------------------------------------
def load_binary_stl_vertices(path: str) -> np.ndarray:
    """Unique vertices (float64, [n,3]) of a binary STL."""
    with open(path, "rb") as f:
        data = f.read()
    (ntri,) = struct.unpack_from("<I", data, 80)
    if len(data) < 84 + 50 * ntri:
        raise ValueError(f"{path}: truncated binary STL")
    rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("attr", "<u2")])
    tris = np.frombuffer(data, dtype=rec, count=ntri, offset=84)
    verts = tris["v"].reshape(-1, 3).astype(np.float64)
    return np.unique(verts, axis=0)


def convex_hull_vertices(points: np.ndarray) -> np.ndarray:
    from scipy.spatial import ConvexHull  # lazy: only needed when compiling from meshes
    hull = ConvexHull(points)
    return points[np.sort(hull.vertices)]


def farthest_point_sample(points: np.ndarray, k: int) -> np.ndarray:
    """Deterministic farthest-point subset of size <= k (first pick: lowest z, then x, y)."""
    n = len(points)
    if n <= k:
        return points
    order = np.lexsort((points[:, 1], points[:, 0], points[:, 2]))
    chosen = [int(order[0])]
    dist = np.linalg.norm(points - points[chosen[0]], axis=1)
    for _ in range(k - 1):
        nxt = int(np.argmax(dist))
        chosen.append(nxt)
        dist = np.minimum(dist, np.linalg.norm(points - points[nxt], axis=1))
    return points[np.sort(chosen)]


def resolve_mesh(filename: str, urdf_path: str) -> str:
    """``package://<model>/meshes/x.STL`` -> path next to the URDF."""
    if filename.startswith("package://"):
        rel = filename[len("package://"):].split("/", 1)[1]
        return os.path.join(os.path.dirname(urdf_path), rel)
    if os.path.isabs(filename):
        return filename
    return os.path.join(os.path.dirname(urdf_path), filename)


# ----------------------------------------------------------------------------
# chain compilation
# ----------------------------------------------------------------------------
def _axis_index(axis: np.ndarray) -> int:
    for i in range(3):
        e = np.zeros(3)
        e[i] = 1.0
        if np.allclose(axis, e, atol=1e-12):
            return i
    raise ValueError(f"joint axis {axis} is not +x/+y/+z; unsupported by the chain kernel")


def compile_urdf(urdf_path: str,
                 actuated: Sequence[str] = ("hip_joint", "knee_joint"),
                 max_torque: Sequence[float] = (2.5, 2.5),
                 gravity_z: float = -9.80665,
                 default_mu: float = 1.0,
                 max_cand_per_link: int = 32,
                 with_meshes: bool = True) -> dict:
    """Compile a URDF into the flat chain description (a plain dict of lists/numbers).

    The result mirrors ``Os2rModel`` field by field and additionally records the
    joint and body names so host code can address joints by name as ScenarIO does.
    """
    urdf = parse_urdf(urdf_path)
    links, joints = urdf["links"], urdf["joints"]
    by_parent: Dict[str, List[dict]] = {}
    for j in joints:
        by_parent.setdefault(j["parent"], []).append(j)
    children = {j["child"] for j in joints}
    roots = [n for n in links if n not in children]
    if len(roots) != 1:
        raise ValueError(f"expected one root link, found {roots}")
    root = roots[0]

    dof_names: List[str] = []
    bodies: List[dict] = []          # one per dof: lists of (mass, com, inertia) and points
    rfix, rpos, axes, damping, friction = [], [], [], [], []

    # pose of the current link frame in the current body frame (or world, before dof 0)
    r_acc, p_acc = np.eye(3), np.zeros(3)
    cur_link = root
    cur_body: Optional[dict] = None  # None: welded to the world (static)
    while True:
        link = links[cur_link]
        if cur_body is not None:
            if link["mass"] > 0.0:
                com_b = p_acc + r_acc @ link["com"]
                in_b = r_acc @ link["inertia"] @ r_acc.T
                cur_body["parts"].append((link["mass"], com_b, in_b, link["name"]))
            if with_meshes:
                for col in link["collisions"]:
                    mesh_path = resolve_mesh(col["mesh"], urdf_path)
                    hull = convex_hull_vertices(load_binary_stl_vertices(mesh_path))
                    pts_link = (col["rot"] @ hull.T).T + col["xyz"]
                    pts_body = (r_acc @ pts_link.T).T + p_acc
                    cur_body["points"].append((link["name"], pts_body))
        outs = by_parent.get(cur_link, [])
        if not outs:
            break
        if len(outs) > 1:
            raise ValueError(f"link {cur_link} has {len(outs)} children; a serial chain is required")
        j = outs[0]
        if j["type"] == "fixed":
            p_acc = p_acc + r_acc @ j["xyz"]
            r_acc = r_acc @ j["rot"]
        elif j["type"] in ("continuous", "revolute"):
            if len(dof_names) == MAX_DOF:
                raise ValueError("more than %d movable joints" % MAX_DOF)
            dof_names.append(j["name"])
            axes.append(_axis_index(j["axis"]))
            rfix.append(r_acc @ j["rot"])
            rpos.append(p_acc + r_acc @ j["xyz"])
            damping.append(j["damping"])
            friction.append(j["friction"])
            cur_body = {"parts": [], "points": [], "links": []}
            bodies.append(cur_body)
            r_acc, p_acc = np.eye(3), np.zeros(3)
        else:
            raise ValueError(f"joint type {j['type']!r} unsupported")
        if cur_body is not None:
            cur_body["links"].append(j["child"])
        cur_link = j["child"]

    nq = len(dof_names)
    if nq < 1:
        raise ValueError("no movable joints")

    mass, com, icom = [], [], []
    for b in bodies:
        m = sum(p[0] for p in b["parts"])
        if m <= 0.0:
            raise ValueError("body without mass: " + ",".join(b["links"]))
        c = sum(p[0] * p[1] for p in b["parts"]) / m
        ic = np.zeros((3, 3))
        for pm, pc, pi, _ in b["parts"]:
            d = pc - c
            ic += pi + pm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        mass.append(m)
        com.append(c)
        icom.append(ic)

    # ---- ground reachability pruning ------------------------------------------------
    # World height of every joint origin is constant up to (and including) the first
    # joint whose axis is not the world vertical; downstream points can drop at most by
    # their path length below that pivot.
    rw, ow = np.eye(3), np.zeros(3)
    pivot_dof = None
    heights_const = []
    for i in range(nq):
        ow = ow + rw @ rpos[i]
        rw_fix = rw @ rfix[i]
        e = np.zeros(3)
        e[axes[i]] = 1.0
        aw = rw_fix @ e
        heights_const.append(ow[2])
        if pivot_dof is None and not np.allclose(np.abs(aw), [0, 0, 1], atol=1e-9):
            pivot_dof = i
            pivot_height = ow[2]
            break
        rw = rw_fix  # rotation about the vertical keeps heights; use q=0

    cand_body: List[int] = []
    cand_p: List[np.ndarray] = []
    cand_link: List[str] = []
    for bi, b in enumerate(bodies):
        for lname, pts in b["points"]:
            keep = pts
            if pivot_dof is None or bi < pivot_dof:
                # body only spins about the vertical: constant heights; never below z=0
                # unless the model is built underground.  (central_pivot_link, whose
                # bottom face sits exactly on z=0, lands here and is excluded.)
                keep = pts[:0]
            else:
                reach = np.linalg.norm(pts, axis=1)
                for k in range(pivot_dof + 1, bi + 1):
                    reach = reach + np.linalg.norm(rpos[k])
                keep = pts[pivot_height - reach < 0.0]
            if len(keep) == 0:
                continue
            keep = farthest_point_sample(keep, max_cand_per_link)
            for p in keep:
                cand_body.append(bi)
                cand_p.append(p)
                cand_link.append(lname)
    if len(cand_body) > MAX_CAND:
        raise ValueError(f"{len(cand_body)} contact candidates exceed OS2R_MAX_CAND={MAX_CAND}")

    act_dof = []
    for name in actuated:
        if name not in dof_names:
            raise ValueError(f"actuated joint {name!r} is not a movable joint of {urdf_path}")
        act_dof.append(dof_names.index(name))

    def sym6(m):
        return [m[0, 0], m[0, 1], m[0, 2], m[1, 1], m[1, 2], m[2, 2]]

    return {
        "name": urdf["name"],
        "nq": nq,
        "dof_names": dof_names,
        "body_links": [b["links"] for b in bodies],
        "axis": [int(a) for a in axes],
        "rfix": [np.asarray(r).reshape(9).tolist() for r in rfix],
        "rpos": [np.asarray(r).tolist() for r in rpos],
        "mass": [float(m) for m in mass],
        "com": [np.asarray(c).tolist() for c in com],
        "icom": [sym6(np.asarray(i)) for i in icom],
        "damping": [float(d) for d in damping],
        "friction": [float(f) for f in friction],
        "mu": [float(default_mu)] * nq,
        "act_dof": act_dof,
        "max_torque": [float(t) for t in max_torque],
        "gravity_z": float(gravity_z),
        "ncand": len(cand_body),
        "cand_body": cand_body,
        "cand_p": [np.asarray(p).tolist() for p in cand_p],
        "cand_link": cand_link,
    }
