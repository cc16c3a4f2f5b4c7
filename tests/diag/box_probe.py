#!/usr/bin/env python3
"""Round 5 (laboratory): how good is the friction box?  The tangential bounds of a contact are mu x the normal impulse that phase 1
leaves (pgs_normal_iters sweeps of the normal and joint-friction rows, from zero).  For the steady state of a bench workload this
prints, per variant, sum |lambda_n(box) - lambda_n(ref)| / sum lambda_n(ref) against two references: the converged normal-only
solve (200 more sweeps of phase 1) and the normal impulses the iteration ends with (the coupled solution).
  python tests/diag/box_probe.py [--workload C4] [--envs 4096] name:warm_p0=1,p1=1 ...        (switches as in r5_rounds.py)"""
import argparse, ctypes, copy, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench                          # noqa: E402
from oracle import oracle_py as O     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C4"); ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--preroll", type=int, default=600); ap.add_argument("--steps", type=int, default=5)
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
L = O.use_laboratory()
L.orc_debug_box_stat.argtypes = [ctypes.POINTER(ctypes.c_double)]


class A:
    workload = a.workload; envs_per_gpu = a.envs; dtype = "f64"; seed = 42
    pgs_iters = None; pgs_normal_iters = None; pgs_tol = None; pgs_exact = None; runtime_model = False


cfg, _, _ = bench.build_config(A, 0, 1)
base = O.OracleSim(cfg, threads=os.cpu_count() or 1)
for _ in range(a.preroll):
    base.step(None)
for v in a.variants:
    name, _, rest = v.partition(":")
    st = dict(kv.split("=") for kv in filter(None, rest.split(",")))
    cfg2 = copy.copy(cfg); cfg2.pgs_normal_iters = int(st.get("p1", 2))
    o = O.OracleSim(cfg2, threads=os.cpu_count() or 1)
    o.set_state(*base.get_state()); o.set_solver_state(*base.get_solver_state())
    for w in (0, 1):
        o.set_action_history(w, base.get_action_history(w))
    for f in range(5):
        o.set_params(f, base.get_params(f))
    o.set_episode_info(*base.episode_info()); o.step_count = base.step_count
    L.orc_set_experimental_warm_p0(int(st.get("warm_p0", 0)))
    for _ in range(3):                 # (the variant's own solver state settles)
        o.step(None)
    L.orc_set_experimental_box_probe(1)
    for _ in range(a.steps):
        o.step(None)
    out = (ctypes.c_double * 4)()
    L.orc_debug_box_stat(out)
    L.orc_set_experimental_box_probe(0); L.orc_set_experimental_warm_p0(0)
    print(f"{name:10s} {st}: against the converged normal-only solve {out[0] / out[1]:.2e}; against the iteration's final normal impulses {out[2] / out[3]:.2e}")
    o.close()
