"""Round 5: what fraction of the contact candidates that a wave scans has a non-zero weight for ANY of its 64
lanes?  Oracle states of the bench workload in its stationary regime; a body is scanned by a wave when some lane's bounding sphere
reaches the contact band (what the kernel's ballot tests).  python tests/diag/cand_void.py [C4|C3|V1] [envs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, bench
from oracle import oracle_py as O
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
class A:
    workload = wl; envs_per_gpu = n; dtype = "f64"; seed = 42
    pgs_iters = None; pgs_normal_iters = None; pgs_tol = None; pgs_exact = None; runtime_model = False
cfg, model, _ = bench.build_config(A, 0, 1)
o = O.OracleSim(cfg, threads=os.cpu_count() or 1)
for _ in range(600):
    o.step(None)
q, qd = o.get_state()
md = cfg.model
nq, nc = md.nq, md.ncand
cand_p = np.array([[md.cand_p[k][j] for j in range(3)] for k in range(nc)])
cand_b = np.array([md.cand_body[k] for k in range(nc)])
center = np.array([[md.cand_center[b][j] for j in range(3)] for b in range(nq)])
radius = np.array([md.cand_radius[b] for b in range(nq)])
margin = cfg.contact_margin
Z = np.zeros((n, nc)); near = np.zeros((n, nq), dtype=bool)
for e in range(n):
    _, _, rw, ow = O.dynamics(md, q[:, e], qd[:, e], np.zeros(nq))
    for b in range(nq):
        ks = np.nonzero(cand_b == b)[0]
        if len(ks) == 0:
            continue
        Z[e, ks] = cand_p[ks] @ rw[b][2] + ow[b][2]
        near[e, b] = center[b] @ rw[b][2] + ow[b][2] - radius[b] * 1.000001 < margin
W = n // 64
live = (Z < margin).reshape(W, 64, nc)
nearw = near.reshape(W, 64, nq).any(axis=1)               # [W, nq] the wave scans body b
scanned = nearw[:, cand_b]                                # [W, nc]
any_lane = live.any(axis=1)                               # candidate has a weight for some lane of the wave
print(f"{wl}, {n} envs = {W} waves; margin {margin}")
print(f"candidates scanned per wave: {scanned.sum(1).mean():.1f} of {nc}; with a weight for some lane: {(any_lane & scanned).sum(1).mean():.1f} "
      f"({(any_lane & scanned).sum() / scanned.sum():.3f}); for a lane on average: {(live & scanned[:, None, :]).sum() / (scanned.sum() * 64):.4f}")
for b in sorted(set(cand_b)):
    ks = cand_b == b
    s = scanned[:, ks]
    if s.sum() == 0:
        print(f"  body {b}: {ks.sum()} candidates, never scanned"); continue
    a = (any_lane[:, ks] & s)
    print(f"  body {b}: {ks.sum()} candidates; scanned by {nearw[:, b].mean():.2f} of the waves; live for some lane: {a.sum() / s.sum():.3f}; "
          f"per candidate (share of scanning waves in which it is live): min {np.min(a.sum(0) / np.maximum(s.sum(0), 1)):.2f} max {np.max(a.sum(0) / np.maximum(s.sum(0), 1)):.2f}")
if os.environ.get("CAND_DETAIL"):
    np.set_printoptions(precision=2, suppress=True, linewidth=200)
    for b in sorted(set(cand_b)):
        ks = np.nonzero(cand_b == b)[0]
        s = scanned[:, ks]
        if s.sum() == 0: continue
        a = (any_lane[:, ks] & s).sum(0) / np.maximum(s.sum(0), 1)
        print(f"body {b}: live share per candidate (table order):"); print(a)
        print("  coordinates:"); print(cand_p[ks].T)

# ---- what clusters with their own boxes (axis-aligned in the body frame) would scan: per heuristic ----
def box_near(ks, e_rw2, e_ow2):
    c = 0.5 * (cand_p[ks].max(0) + cand_p[ks].min(0)); h = 0.5 * (cand_p[ks].max(0) - cand_p[ks].min(0)) * 1.000001 + 1e-12
    return (e_rw2 @ c + e_ow2 - np.abs(e_rw2) @ h) < margin


def evaluate(name, clusters_of_body):
    """clusters_of_body: {b: [index arrays]}"""
    RW2 = np.zeros((n, nq, 3)); OW2 = np.zeros((n, nq))
    for e in range(n):
        _, _, rw, ow = O.dynamics(md, q[:, e], qd[:, e], np.zeros(nq))
        RW2[e] = rw[:, 2, :]; OW2[e] = ow[:, 2]
    tot_scan = tot_tests = 0.0
    for b, cls in clusters_of_body.items():
        body_near = nearw[:, b]
        for ks in cls:
            nr_ = np.array([box_near(ks, RW2[e, b], OW2[e, b]) for e in range(n)]).reshape(W, 64).any(axis=1) & body_near
            tot_scan += nr_.sum() * len(ks)
        tot_tests += body_near.sum() * (len(cls) if len(cls) > 1 else 0)
    print(f"  {name:34s}: candidates scanned per wave {tot_scan / W:6.1f}   cluster tests per wave {tot_tests / W:4.1f}")


def split_axes(ks, axes_):
    out = [ks]
    for a in axes_:
        nxt = []
        for g in out:
            mid = 0.5 * (cand_p[g, a].max() + cand_p[g, a].min())
            lo_, hi_ = g[cand_p[g, a] <= mid], g[cand_p[g, a] > mid]
            nxt += [x for x in (lo_, hi_) if len(x)]
        out = nxt
    return out


if os.environ.get("CAND_CLUSTERS"):
    bodies_ = [b for b in sorted(set(cand_b)) if nearw[:, b].any()]
    idx = {b: np.nonzero(cand_b == b)[0] for b in bodies_}
    ext = {b: cand_p[idx[b]].max(0) - cand_p[idx[b]].min(0) for b in bodies_}
    print("clusterings (boxes axis-aligned in the body frame):")
    evaluate("one cluster per body (box)", {b: [idx[b]] for b in bodies_})
    evaluate("halves, thinnest axis", {b: split_axes(idx[b], [int(np.argsort(ext[b])[0])]) for b in bodies_})
    evaluate("halves, longest axis", {b: split_axes(idx[b], [int(np.argsort(ext[b])[2])]) for b in bodies_})
    evaluate("quadrants, two thinnest axes", {b: split_axes(idx[b], list(np.argsort(ext[b])[:2])) for b in bodies_})
    evaluate("quadrants, thinnest + longest", {b: split_axes(idx[b], [int(np.argsort(ext[b])[0]), int(np.argsort(ext[b])[2])]) for b in bodies_})
    evaluate("octants", {b: split_axes(idx[b], [0, 1, 2]) for b in bodies_})
if os.environ.get("CAND_CLUSTERS") and wl in ("C4", "C3"):
    i3, i4 = idx[3], idx[4]
    live3 = i3[11:19]; rest3 = np.setdiff1d(i3, live3)
    l4 = np.array([0, 1, 2, 13, 14, 15, 16, 17, 18, 19, 20, 29, 30, 31]); live4 = i4[l4]; rest4 = np.setdiff1d(i4, live4)
    evaluate("oracle-informed: live set + rest", {2: [idx[2]], 3: [live3, rest3], 4: [live4, rest4]})
    evaluate("  ... rest split along its longest axis", {2: [idx[2]], 3: [live3] + split_axes(rest3, [2]), 4: [live4] + split_axes(rest4, [1])})
