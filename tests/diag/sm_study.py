#!/usr/bin/env python3
"""Prototype of the exact finish of the boxed LCP (round 3): projected Gauss-Seidel + subspace minimisation.

  python tests/diag/sm_study.py [--regime bench|balancing] [--k0 4] [--k1 1] [--rounds 3] [--eps 1e-10]

Works on the exported problems of tests/diag/solver_study.py (cached under /tmp), in the whitened coordinates of the
kernel (G = J Lc, y = Lc^-1 v, Minv = Lc Lc^T).  After `k0` ordinary sweeps the rows strictly inside their box (the free
set F) are solved exactly:  the velocity change of the free rows is the minimum-norm d with G_F (y + d) = t_F, i.e.
    (eps I + G_F^T G_F) d = -G_F^T w_F,      w_F = G_F y - t_F,      impulses  mu_r = -(w_r + g_r.d) / eps
-- a 5 x 5 SPD system whatever the number of free rows (8 sticking rows in 5 dof are no special case), regularised by
eps * trace so that d stays in the range of G_F^T.  `k1` sweeps follow (the last one measures), and the stopping rule
decides whether another round is needed.
"""
import argparse
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "diag"))
import lcp_ref                       # noqa: E402
from oracle import oracle_py as O    # noqa: E402


def problems(regime):
    path = f"/tmp/os2r_problems_{regime}.pkl"
    if os.path.exists(path):
        return pickle.load(open(path, "rb"))
    import solver_study
    O.build()
    ps = solver_study.collect(regime)
    pickle.dump(ps, open(path, "wb"))
    return ps


class Prob:
    def __init__(self, p):
        self.p = p
        self.Lc = np.linalg.cholesky(p["minv"])
        self.G = p["J"] @ self.Lc
        self.d = np.einsum("rk,rk->r", self.G, self.G)
        self.t = p["target"]
        self.kind = p["kind"]
        self.nrow = p["normal_row"]
        self.y0 = np.linalg.solve(self.Lc, p["vstar"])
        self.nr = len(self.t)

    def v(self, y):
        return self.Lc @ y


def row(P, r, y, lam, lo, hi):
    if not P.d[r] > 0:
        return 0.0
    res = P.G[r] @ y - P.t[r]
    new = min(max(lam[r] - res / P.d[r], lo), hi)
    dl = new - lam[r]
    lam[r] = new
    y += P.G[r] * dl
    return abs(res * dl)


def sweep(P, y, lam, lo, hi):
    return sum(row(P, r, y, lam, lo[r], hi[r]) for r in range(P.nr))


def phase1(P, sweeps=3):
    y, lam = P.y0.copy(), np.zeros(P.nr)
    for _ in range(sweeps):
        for r in range(P.nr):
            if P.kind[r] == 0:
                row(P, r, y, lam, 0.0, np.inf)
            elif P.kind[r] == 2:
                row(P, r, y, lam, -P.p["bound"][r], P.p["bound"][r])
    box = P.p["bound"].copy()
    for r in range(P.nr):
        if P.kind[r] == 1:
            box[r] = P.p["bound"][r] * lam[P.nrow[r]]
    lo = np.where(P.kind == 0, 0.0, -box)
    hi = np.where(P.kind == 0, np.inf, box)
    return y, lam, lo, hi


def sm_step(P, y, lam, lo, hi, eps_rel, truncate):
    F = (P.d > 0) & (lam > lo) & (lam < hi)
    if not F.any():
        return 0
    GF = P.G[F]
    S = GF.T @ GF
    eps = eps_rel * np.trace(S)
    w = GF @ y - P.t[F]
    h = -GF.T @ w
    d = np.linalg.solve(S + eps * np.eye(len(y)), h)
    mu = -(w + GF @ d) / eps
    a = 1.0
    if truncate:
        lf, lof, hif = lam[F], lo[F], hi[F]
        with np.errstate(divide="ignore", invalid="ignore"):
            lim = np.where(mu > 0, (hif - lf) / mu, np.where(mu < 0, (lof - lf) / mu, np.inf))
        a = min(1.0, lim.min())
    y += a * d
    lam[F] += a * mu
    return int(F.sum())


def solve(P, k0, k1, rounds, eps_rel, tol, truncate, cap=None):
    """-> (y, sweeps run, sm steps run)"""
    y, lam, lo, hi = phase1(P)
    ns = 0
    for k in range(k0):
        E = sweep(P, y, lam, lo, hi); ns += 1
    if E <= tol:
        return y, ns, 0
    for rd in range(rounds):
        sm_step(P, y, lam, lo, hi, eps_rel, truncate)
        for k in range(k1):
            E = sweep(P, y, lam, lo, hi); ns += 1
        if E <= tol:
            return y, ns, rd + 1
    return y, ns, rounds


def reference(P):
    """converged / exact velocity of the fixed-box problem: long Gauss-Seidel, enumeration where that has not converged"""
    y, lam, lo, hi = phase1(P)
    for it in range(3000):
        E = sweep(P, y, lam, lo, hi)
        if E == 0.0:
            break
    pc = dict(P.p); pc["box"] = np.where(P.kind == 0, np.inf, hi)
    A, c, lo_, hi_ = lcp_ref.lcp_matrices(pc)
    res = lcp_ref.kkt_residual(A, c, lo_, hi_, lam)
    if res > 1e-13 and P.nr <= 11:
        lx, rx = lcp_ref.enumerate_exact(A, c, lo_, hi_, tol=1e-12)
        if rx < res:
            return lcp_ref.velocity(pc, lx), rx
    return P.v(y), res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--regime", default="bench")
    ap.add_argument("--k0", type=int, default=4)
    ap.add_argument("--k1", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--eps", type=float, default=1e-10)
    ap.add_argument("--tol", type=float, default=1e-24)
    ap.add_argument("--truncate", type=int, default=0)
    ap.add_argument("--max", type=int, default=100000)
    a = ap.parse_args()
    ps = [Prob(p) for p in problems(a.regime) if p["nr"] > 5][: a.max]
    refpath = f"/tmp/os2r_problems_{a.regime}_ref.pkl"
    if os.path.exists(refpath):
        refs = pickle.load(open(refpath, "rb"))
    else:
        refs = [reference(P) for P in ps]
        pickle.dump(refs, open(refpath, "wb"))
    rres = np.array([r[1] for r in refs])
    print(f"{len(ps)} problems with contact rows; reference optimality residual p50 {np.median(rres):.1e} p99 {np.percentile(rres, 99):.1e} max {rres.max():.1e}")
    errs, sweeps, sms = [], [], []
    for P, (vx, _) in zip(ps, refs):
        y, ns, nsm = solve(P, a.k0, a.k1, a.rounds, a.eps, a.tol, a.truncate)
        errs.append(np.abs(P.v(y) - vx).max() / max(1.0, np.abs(vx).max()))
        sweeps.append(ns); sms.append(nsm)
    errs, sweeps, sms = map(np.array, (errs, sweeps, sms))
    ok = rres < 1e-9
    print(f"k0={a.k0} k1={a.k1} rounds={a.rounds} eps={a.eps:g} truncate={a.truncate}: error vs reference p50 {np.median(errs[ok]):.1e} p90 {np.percentile(errs[ok], 90):.1e} "
          f"p99 {np.percentile(errs[ok], 99):.1e} max {errs[ok].max():.1e}; sm steps histogram {np.bincount(sms)}; sweeps mean {sweeps.mean():.2f} max {sweeps.max()}")
    # plain 20 sweeps for comparison
    e20 = []
    for P, (vx, _) in zip(ps, refs):
        y, lam, lo, hi = phase1(P)
        for _ in range(20):
            sweep(P, y, lam, lo, hi)
        e20.append(np.abs(P.v(y) - vx).max() / max(1.0, np.abs(vx).max()))
    e20 = np.array(e20)
    print(f"plain 3 + 20: p50 {np.median(e20[ok]):.1e} p90 {np.percentile(e20[ok], 90):.1e} p99 {np.percentile(e20[ok], 99):.1e} max {e20[ok].max():.1e}")


if __name__ == "__main__":
    main()
