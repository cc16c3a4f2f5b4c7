"""Independent numpy checks of the contact half of the oracle (tests/test_oracle_contact.py).

Nothing here shares code with oracle/os2r_oracle.c: the boxed LCP of one physics iteration is taken as data
(oracle_py.contact_problem) and
  * restated: the two-phase projected Gauss-Seidel of DESIGN.md 3.2 step 6, with its stopping rule;
  * solved exactly: enumeration of the active sets of the fixed-box problem (a convex QP: the first
    assignment that meets the optimality conditions IS the solution; A is only positive semi-definite, so the
    free block is solved in the least-squares sense and the velocity, which is unique, is compared);
  * certified: the optimality (KKT) residual of any candidate solution, in whitened units.
"""
import itertools

import numpy as np


def lcp_matrices(p):
    """A = J Minv J^T, c = J v* - target, and the fixed box [lo, hi] of phase 2."""
    J, minv = p["J"], p["minv"]
    A = J @ minv @ J.T
    c = J @ p["vstar"] - p["target"]
    lo = np.where(p["kind"] == 0, 0.0, -p["box"])
    hi = np.where(p["kind"] == 0, np.inf, p["box"])
    return A, c, lo, hi


def velocity(p, lam):
    return p["vstar"] + p["minv"] @ (p["J"].T @ lam)


def kkt_residual(A, c, lo, hi, lam):
    """max over rows of the violation of (lam = lo: w >= 0 | lam = hi: w <= 0 | else w = 0), w = A lam + c,
    each row scaled by sqrt(A_rr) (the row's norm in the whitened metric); rows with lo == hi carry no condition."""
    w = A @ lam + c
    scale = np.sqrt(np.maximum(np.diag(A), 1e-300))
    span = np.maximum(np.abs(lam), 1e-300)
    at_lo = lam <= lo + 1e-12 * span
    at_hi = lam >= hi - 1e-12 * span
    v = np.where(at_lo & at_hi, 0.0, np.where(at_lo, np.maximum(-w, 0), np.where(at_hi, np.maximum(w, 0), np.abs(w))))
    return float((v / scale).max())


def enumerate_exact(A, c, lo, hi, tol=1e-9):
    """-> (lam, residual) of the first active-set assignment that satisfies the optimality conditions."""
    nr = len(c)
    opts = []
    for r in range(nr):
        if hi[r] - lo[r] <= 0: opts.append("P")            # pinned: both bounds coincide
        elif np.isinf(hi[r]): opts.append("FL")
        else: opts.append("FLU")
    scale = np.sqrt(np.maximum(np.diag(A), 1e-300))
    best = None
    for st in itertools.product(*opts):
        st = np.array(st)
        F = st == "F"
        lam = np.where(st == "U", hi, lo).astype(float)
        lam[F] = 0.0
        if F.any():
            rhs = -(c[F] + A[np.ix_(F, ~F)] @ lam[~F])
            lam[F] = np.linalg.lstsq(A[np.ix_(F, F)], rhs, rcond=1e-13)[0]
        w = A @ lam + c
        v = np.where(st == "P", 0.0, np.where(st == "L", np.maximum(-w, 0) / scale, np.where(st == "U", np.maximum(w, 0) / scale,
                     np.maximum(np.abs(w) / scale, np.maximum(np.maximum(lo - lam, lam - np.where(np.isinf(hi), lam, hi)), 0) * scale))))
        res = float(v.max())
        if best is None or res < best[1]:
            best = (lam.copy(), res)
        if res < tol:
            break
    return best


def pgs_two_phase(p, normal_iters=3, iters=20, tol=1e-24, group=4):
    """The specification's solver on the exported rows: -> (v, lam, box, sweeps of phase 2 actually run)."""
    J, minv, t, kind, nrow = p["J"], p["minv"], p["target"], p["kind"], p["normal_row"]
    nr = len(t)
    T = J @ minv                      # T[r] = Minv J_r (Minv symmetric)
    d = np.einsum("rj,rj->r", J, T)
    v = p["vstar"].copy()
    lam = np.zeros(nr)
    box = p["bound"].copy()
    ran = 0
    for phase in (0, 1):
        if phase == 1:
            for r in range(nr):
                if kind[r] == 1:
                    box[r] = p["bound"][r] * lam[nrow[r]]
        for it in range(normal_iters if phase == 0 else iters):
            moved = 0.0
            for r in range(nr):
                if not d[r] > 0 or (phase == 0 and kind[r] == 1):
                    continue
                res = J[r] @ v - t[r]
                new = lam[r] - res / d[r]
                if kind[r] == 0:
                    new = max(new, 0.0)
                else:
                    new = min(max(new, -box[r]), box[r])
                dl = new - lam[r]
                lam[r] = new
                moved += abs(res) * abs(dl)
                v += T[r] * dl
            if phase == 1:
                ran += 1
                if (it + 1) % group == 0 and it + 1 < iters and moved <= tol:
                    break
    return v, lam, box, ran


def pgs_exact_finish(p, normal_iters=3, iters=None, tol=1e-24, exact=12, first=None, eps_rel=1e-6, prox=2, snap=1e-12, incons=1e-4,
                     small=0, small_pivot=1e-8, equil=True):
    """The specification's solver with the exact finish, restated on the
    exported rows: phase 1 as in pgs_two_phase; phase 2 = `first` sweeps, then -- while the last sweep moved more than
    `tol` -- exact solves of the free rows (repeated while a bound cuts the step short, `exact` at most) each followed
    by one sweep; `iters` bounds the sweeps.  From the second solve of the call on, a solve that is not cut and leaves
    more than `incons` of the squared residual it found on its free rows (an inconsistent free set) goes on along its
    multipliers to the first bound and counts as cut.  `small` > 0 (studies; not the specification): a free set of at most that
    many rows (two at most here) is solved in the dual, (G_F G_F^T) mu = -w, unless its second pivot is below `small_pivot`.
    `equil` (the specification since round 5): every free row enters the regularised solve with the weight 1 / |g_r|^2.
    -> (v, lam, box, sweeps of phase 2, exact solves)"""
    J, minv, t, kind, nrow = p["J"], p["minv"], p["target"], p["kind"], p["normal_row"]
    nr, n = J.shape
    if first is None:
        first = 6 if n >= 5 else 4     # sweeps before the first check: per number of dof
    if iters is None:
        iters = first + 8
    Lc = np.linalg.cholesky(minv)
    G = J @ Lc                         # rows in the whitened coordinates y = Lc^-1 v
    d = np.einsum("rk,rk->r", G, G)
    y = np.linalg.solve(Lc, p["vstar"])
    lam = np.zeros(nr)
    box = p["bound"].copy()

    def sweep(rows, lo, hi):
        moved = 0.0
        for r in rows:
            if not d[r] > 0:
                continue
            res = G[r] @ y - t[r]
            new = min(max(lam[r] - res / d[r], lo[r]), hi[r])
            dl = new - lam[r]
            lam[r] = new
            moved += abs(res) * abs(dl)
            y[:] += G[r] * dl
        return moved

    lo = np.where(kind == 0, 0.0, -box)
    hi = np.where(kind == 0, np.inf, box)
    for _ in range(normal_iters):
        sweep([r for r in range(nr) if kind[r] != 1], lo, hi)
    for r in range(nr):
        if kind[r] == 1:
            box[r] = p["bound"][r] * lam[nrow[r]]
    lo = np.where(kind == 0, 0.0, -box)
    hi = np.where(kind == 0, np.inf, box)
    ran = solves = 0
    for it in range(iters):
        if it >= first and solves < exact:
            blocked = True
            while blocked and solves < exact:
                solves += 1
                F = (d > 0) & (lam > lo) & (lam < hi)
                wgt = 1.0 / d[F] if equil else np.ones(int(F.sum()))     # equil: every free row is taken with weight 1 / |g_r|^2
                S = (G[F] * wgt[:, None]).T @ G[F]
                tr = np.trace(S)
                if not tr > 0:
                    blocked = False
                    break
                eps = eps_rel * tr
                w = G[F] @ y - t[F]
                A2 = G[F] @ G[F].T
                dual = 0 < F.sum() <= small and (F.sum() == 1 or A2[1, 1] - A2[0, 1] ** 2 / A2[0, 0] > small_pivot * A2[1, 1])
                if dual:
                    mu = np.linalg.solve(A2, -w)
                    full = lam[F] + mu
                    blocked = bool(((full < lo[F]) | (full > hi[F])).any())
                    alpha = 1.0
                    if blocked:
                        with np.errstate(divide="ignore", invalid="ignore"):
                            lim = np.where(mu > 0, (hi[F] - lam[F]) / mu, np.where(mu < 0, (lo[F] - lam[F]) / mu, np.inf))
                        alpha = min(1.0, lim.min())
                    y += alpha * (G[F].T @ mu)
                    for j, m in zip(np.nonzero(F)[0], mu):
                        nl = lam[j] + alpha * m
                        if blocked:
                            if m > 0 and np.isfinite(hi[j]) and hi[j] - nl <= snap * (hi[j] - lam[j]):
                                nl = hi[j]
                            if m < 0 and nl - lo[j] <= snap * (lam[j] - lo[j]):
                                nl = lo[j]
                        lam[j] = min(max(nl, lo[j]), hi[j])
                    continue
                h = -G[F].T @ (wgt * w)
                A = S + eps * np.eye(n)
                dk = np.zeros(n)
                ds = np.zeros(n)
                for k in range(prox):
                    dk = np.linalg.solve(A, h + eps * dk)
                    ds += dk
                mu = -wgt * (prox * w + G[F] @ ds) / eps
                full = lam[F] + mu
                blocked = bool(((full < lo[F]) | (full > hi[F])).any())      # the full step leaves a box: cut it
                left = w + G[F] @ dk
                on = solves > 1 and not blocked and left @ left > incons * (w @ w)
                alpha = 1.0
                if blocked or on:
                    with np.errstate(divide="ignore", invalid="ignore"):
                        lim = np.where(mu > 0, (hi[F] - lam[F]) / mu, np.where(mu < 0, (lo[F] - lam[F]) / mu, np.inf))
                    alpha = min(1.0, lim.min()) if blocked else lim.min()
                    if on:                           # inconsistent free set: on to the first bound, however far
                        blocked = bool(np.isfinite(alpha))
                        alpha = alpha if blocked else 1.0
                y += alpha * dk
                idx = np.nonzero(F)[0]
                for j, m in zip(idx, mu):
                    nl = lam[j] + alpha * m
                    if blocked:                      # a row the cut step has taken to its bound is set on it
                        if m > 0 and np.isfinite(hi[j]) and hi[j] - nl <= snap * (hi[j] - lam[j]):
                            nl = hi[j]
                        if m < 0 and nl - lo[j] <= snap * (lam[j] - lo[j]):
                            nl = lo[j]
                    lam[j] = min(max(nl, lo[j]), hi[j])
        moved = sweep(range(nr), lo, hi)
        ran += 1
        if it + 1 < iters and it + 1 >= first and moved <= tol:
            break
    return Lc @ y, lam, box, ran, solves
