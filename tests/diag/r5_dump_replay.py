"""Round 5: the hard problems of the exact finish, one by one.  The laboratory oracle dumps every problem that took k or more solves
(ORC_DUMP_SOLVES=<file> ORC_TRACE_SOLVES=k, one JSON object per line); this replays them with the numpy restatement
(tests/lcp_ref.py) and compares with the exact solution by enumeration of the active sets (3^rows assignments: <= 11 rows).
  python tests/diag/r5_dump_replay.py /tmp/dump.jsonl [max problems]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lcp_ref

def load(line):
    d = json.loads(line)
    n, nr = d["n"], d["nr"]
    R = d["rows"]
    p = {"J": np.array([r["J"] for r in R]), "minv": np.array(d["minv"]).reshape(n, n), "target": np.array([r["target"] for r in R]),
         "kind": np.array([r["kind"] for r in R]), "normal_row": np.array([r["normal_row"] for r in R]), "body": np.array([r["body"] for r in R]),
         "bound": np.array([r["bound"] for r in R]), "vstar": np.array(d["vstar"]), "solves": d["solves"]}
    return p

def main():
    lines = open(sys.argv[1]).read().splitlines()
    lim = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    for ln in lines[:lim]:
        p = load(ln)
        nr = len(p["target"])
        v, lam, box, ran, solves = lcp_ref.pgs_exact_finish(p, first=3, iters=14)
        p["box"] = box
        A, c, lo, hi = lcp_ref.lcp_matrices(p)
        res = lcp_ref.kkt_residual(A, c, lo, hi, lam)
        desc = "".join("n" if k == 0 else ("t" if k == 1 else "j") for k in p["kind"])
        bodies = sorted(set(p["body"][p["kind"] == 0]))
        line = f"rows {desc} contact bodies {bodies} oracle solves {p['solves']} numpy cold solves {solves} sweeps {ran} kkt residual {res:.1e}"
        if nr <= 11:
            lam_x, res_x = lcp_ref.enumerate_exact(A, c, lo, hi)
            vx = lcp_ref.velocity(p, lam_x)
            st = "".join("F" if lo[r] < lam_x[r] < hi[r] else ("L" if lam_x[r] <= lo[r] else "U") for r in range(nr))
            stn = "".join("F" if lo[r] < lam[r] < hi[r] else ("L" if lam[r] <= lo[r] else "U") for r in range(nr))
            G = p["J"] @ np.linalg.cholesky(p["minv"])
            Fm = np.array([ch == "F" for ch in st])
            rank = np.linalg.matrix_rank(G[Fm], tol=1e-9 * np.abs(G).max()) if Fm.any() else 0
            line += f" | exact set {st} (residual {res_x:.1e}, free rows {Fm.sum()} of rank {rank}) solver ends at {stn}, |v - v_exact| {np.abs(v - vx).max():.1e} of |v| {np.abs(vx).max():.1e}"
        print(line)

if __name__ == "__main__":
    main()
