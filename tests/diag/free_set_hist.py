"""Round 5: the shape of the free set of every exact solve, by the solve's index within its physics iteration (oracle laboratory,
bench workload in its stationary regime):  python tests/diag/free_set_hist.py [C4|C3|V1] [envs] [steps]
Classes: pair = the normal and ONE tangential row of one contact; nn = two normals; one = a single row; ntt = the three rows of one
contact; <=3 = any other set of at most three rows; >=4; '+j' = the same with a joint-friction row among them."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, bench
from oracle import oracle_py as O
wl = sys.argv[1] if len(sys.argv) > 1 else 'C4'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
class A:
    workload = wl; envs_per_gpu = n; dtype = "f64"; seed = 42
    pgs_iters = None; pgs_normal_iters = None; pgs_tol = None; pgs_exact = None; runtime_model = False
cfg, _, _ = bench.build_config(A, 0, 1)
L = O.use_laboratory()
o = O.OracleSim(cfg, threads=os.cpu_count() or 1)
for _ in range(400):
    o.step(None)
L.orc_debug_counter(0, 1)
h = np.zeros((16, 12), dtype=np.int64)
L.orc_debug_free_set_hist(h.ctypes.data_as(ctypes.c_void_p), 1)
for _ in range(steps):
    o.step(None)
L.orc_debug_free_set_hist(h.ctypes.data_as(ctypes.c_void_p), 1)
names = ["pair", "nn", "one", "ntt", "<=3", ">=4"]
names = names + [x + "+j" for x in names]
li = n * steps * int(cfg.substeps)
print(f"{wl}: {n} envs x {steps} env-steps = {li} lane-iterations; solves {h.sum()} ({h.sum() / li:.4f} per lane-iteration)")
print("solve#  " + " ".join(f"{x:>8s}" for x in names) + "    total  share")
for k in range(16):
    if h[k].sum() == 0: continue
    print(f"{k:5d}   " + " ".join(f"{v:8d}" for v in h[k]) + f" {h[k].sum():8d}  {h[k].sum() / h.sum():.4f}")
print("all     " + " ".join(f"{v:8d}" for v in h.sum(0)))
print("share   " + " ".join(f"{v / h.sum():8.4f}" for v in h.sum(0)))
